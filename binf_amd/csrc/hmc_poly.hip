// Fused HMC transition on the example's polynomial posterior for SMALL data
// sets (n_data <= 128, K <= 16 coefficients): the reference's own
// example_script.py shape (K = 4, n_data = 20, 50 leapfrog steps), where the
// per-step tier is launch-bound (~150 launches, 2 ms per transition however
// few the chains).  ONE launch = one HMCSampler.sample() for every chain.
// This file: the entry point binf_hmc_sample_poly_f64 and the ONE-LANE-PER-CHAIN
// kernel it runs under BINF_MODE_LANE_PER_CHAIN (the layout for ~10^5 chains and
// more; by default a chain is a lane group, poly_chain_kernel.hpp).  Replaces
// binf/samplers/hmc.py:92-164 around
//   Posterior.log_prob / gradient        binf/pdf/posteriors.py:147-187
//   Likelihood.log_prob / gradient       binf/pdf/likelihoods.py:141-155
//   ForwardModel / GaussianErrorModel    binf/example/likelihood.py:24-30,54-61
//   GaussianPrior.log_prob               binf/example/priors.py:49-54
// gfx950 (MI355X), wave64.
//
// Mapping: ONE LANE PER CHAIN (theta, p, force in registers), one wave per
// workgroup so that few chains still spread over the CUs; xs / ys are staged
// in LDS and read as broadcasts.  Every sum is evaluated by its lane alone, so
// the energies can follow numpy to the bit:
//   chi^2          Horner in polyval's order, residual squares in np.sum's
//                  order (one pairwise leaf: n_data <= 128),
//   prior, kinetic np.sum's order over K elements,
//   log-prob       the Posterior's component terms added one after the other
//                  in the order the host passes (sorted component names),
// i.e. E_before / E_after are bit-identical to the per-step tier's.  The force
// (theta-gradient of the likelihood only: the Gaussian prior is registered
// non-differentiable, quirk Q4) is a plain FMA dot product -- the per-step
// tier computes it on the MFMA pipe, the reference with BLAS; summation order
// is not reproducible there either -- so trajectories agree to ~1e-15
// relative, not bitwise.
#include "hmc_poly_args.hpp"

namespace binf {

// np.sum of the n <= 128 values f(0), f(1), ... evaluated in order by one lane
// (n is wave-uniform): numpy's pairwise leaf, then the outer "0.0 +".
template <class F>
__device__ inline double np_sum_leaf(F f, int n)
{
    double res;
    if (n < 8) {
        res = -0.0;
        for (int i = 0; i < n; ++i) res = res + f(i);
    } else {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = f(j);
        const int n8 = n & ~7;
        for (int i = 8; i < n8; i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = r[j] + f(i + j);
        }
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (int i = n8; i < n; ++i) res = res + f(i);
    }
    return 0.0 + res;
}

// The same for K <= KMAX values held in registers: f is called with
// compile-time indices only, runtime K (wave-uniform) masks them.
template <int KMAX, class F>
__device__ inline double np_sum_regs(F f, int K)
{
    double res;
    if (KMAX < 8 || K < 8) {
        res = -0.0;
#pragma unroll
        for (int i = 0; i < (KMAX < 7 ? KMAX : 7); ++i) {
            const double n = res + f(i);
            res = (i < K) ? n : res;
        }
    } else {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = f(j);
        const int k8 = K & ~7;
#pragma unroll
        for (int i = 8; i < KMAX; ++i) {
            const double n = r[i & 7] + f(i);
            r[i & 7] = (i < k8) ? n : r[i & 7];
        }
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
        for (int i = 8; i < KMAX; ++i) {
            const double n = res + f(i);
            res = (i >= k8 && i < K) ? n : res;
        }
    }
    return 0.0 + res;
}

template <int KMAX, bool FMA>
__global__ void __launch_bounds__(64) hmc_poly_small_kernel(const PolyHmcArgs a)
{
    __shared__ double sx[128], sy[128];
    const int lane = threadIdx.x;
    const int K = a.K, N = a.N;
    for (int i = lane; i < N; i += 64) {
        sx[i] = a.xs[i];
        sy[i] = a.ys[i];
    }
    __syncthreads();

    const int64_t raw = (int64_t)blockIdx.x * 64 + lane;
    const bool valid = raw < a.C;
    const int64_t c = valid ? raw : a.C - 1;

    double th[KMAX], p[KMAX], old[KMAX], pm[KMAX], pv[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        th[k] = (k < K) ? a.q0[c * K + k] : 0.0;
        p[k] = (k < K) ? a.p0[c * K + k] : 0.0;
        old[k] = th[k];
        pm[k] = (a.prior_means && k < K) ? a.prior_means[k] : 0.0;
        pv[k] = (a.prior_means && k < K) ? a.prior_vars[k] : 1.0;
    }
    const double tau = a.tau_chain ? a.tau_chain[c] : a.tau;
    double dt = a.dt_chain ? a.dt_chain[c] : a.timestep;
    const double uu = a.u[c];
    const double logZ = (double)N * 0.5 * log(tau);               // likelihood.py:55
    const double lp_pre = a.lp_pre ? a.lp_pre[c] : 0.0;
    const double lp_post = a.lp_post ? a.lp_post[c] : 0.0;

    // log posterior, component terms added one after the other in the
    // Posterior's order (posteriors.py:147-151, sorted component names)
    auto log_prob = [&]() {
        auto resid2 = [&](int n) {
            // polyval: Horner, zero-padded above K-1 (exact no-ops)
            const double x = sx[n];
            double v = th[KMAX - 1] + x * 0.0;
#pragma unroll
            for (int k = KMAX - 2; k >= 0; --k) v = th[k] + v * x;
            const double d = v - sy[n];
            return d * d;
        };
        const double chi2 = 1.0 * np_sum_leaf(resid2, N);
        const double lik = -0.5 * chi2 * tau + logZ;              // likelihood.py:56-57
        double pri = 0.0;
        if (a.prior_means) {
            auto term = [&](int k) {
                const double d = th[k] - pm[k];
                return d * d / pv[k];                             // priors.py:52-54
            };
            pri = -0.5 * np_sum_regs<KMAX>(term, K);
        }
        double total = 0.0;
        bool have = false;
        auto add = [&](double t) {
            total = have ? total + t : t;
            have = true;
        };
        if (a.lp_pre) add(lp_pre);
        if (a.prior_means && a.prior_first) add(pri);
        add(lik);
        if (a.prior_means && !a.prior_first) add(pri);
        if (a.lp_post) add(lp_post);
        return total;
    };
    auto kinetic = [&]() {                                        // hmc.py:148,150
        auto sq = [&](int k) { return p[k] * p[k]; };
        return 0.5 * np_sum_regs<KMAX>(sq, K);
    };
    // force = tau * sum_n (polyval(x_n) - y_n) x_n^k             likelihoods.py:148-155
    double g[KMAX];
    auto force = [&]() {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) g[k] = 0.0;
        // four data points per round: their LDS reads and Horner chains overlap (one
        // point at a time the loop waited ~100 cycles for LDS per point, with a single
        // wave per SIMD at the example's batch sizes); each g[k] still receives its
        // terms in the order n = 0, 1, 2, ... -- the same bits as the plain loop
        int n = 0;
        for (; n + 4 <= N; n += 4) {
            double x[4], y[4], v[4], r[4], pw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { x[u] = sx[n + u]; y[u] = sy[n + u]; v[u] = th[KMAX - 1]; }
#pragma unroll
            for (int k = KMAX - 2; k >= 0; --k) {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = __builtin_fma(v[u], x[u], th[k]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { r[u] = (v[u] - y[u]) * tau; pw[u] = 1.0; }
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    g[k] = __builtin_fma(pw[u], r[u], g[k]);
                    pw[u] = pw[u] * x[u];
                }
            }
        }
        for (; n < N; ++n) {
            const double x = sx[n];
            double v = th[KMAX - 1];
#pragma unroll
            for (int k = KMAX - 2; k >= 0; --k) v = __builtin_fma(v, x, th[k]);
            const double r = (v - sy[n]) * tau;
            double pw = 1.0;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                g[k] = __builtin_fma(pw, r, g[k]);
                pw = pw * x;
            }
        }
        // padded coefficients above K-1 stay exactly zero
#pragma unroll
        for (int k = 0; k < KMAX; ++k) g[k] = (k < K) ? g[k] : 0.0;
    };

    const double e_before = -log_prob() + kinetic();             // hmc.py:148
    const double hdt = 0.5 * dt;
    force();                                                      // hmc.py:116
#pragma unroll
    for (int k = 0; k < KMAX; ++k) p[k] = kick<FMA>(p[k], hdt, g[k]);
    for (int l = 0; l < a.nsteps - 1; ++l) {                      // hmc.py:118-120
#pragma unroll
        for (int k = 0; k < KMAX; ++k) th[k] = drift<FMA>(th[k], p[k], dt);
        force();
#pragma unroll
        for (int k = 0; k < KMAX; ++k) p[k] = kick<FMA>(p[k], dt, g[k]);
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) th[k] = drift<FMA>(th[k], p[k], dt);   // hmc.py:122-123
    force();
#pragma unroll
    for (int k = 0; k < KMAX; ++k) p[k] = kick<FMA>(p[k], hdt, g[k]);
    const double e_after = -log_prob() + kinetic();              // hmc.py:150

    double x = -(e_after - e_before);                            // hmc.py:151
    x = (x < -308.0) ? -308.0 : x;
    x = (x > 709.0) ? 709.0 : x;
    const bool acc = uu < exp_clipped_range(x);
    if (!valid) return;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) a.q_out[c * K + k] = acc ? th[k] : old[k];
    a.accepted[c] = acc ? 1 : 0;
    if (a.n_accepted && acc) a.n_accepted[c] += 1;
    if (a.e_before) a.e_before[c] = e_before;
    if (a.e_after) a.e_after[c] = e_after;
    if (a.adapt) a.dt_chain[c] = acc ? dt * a.uprate : dt * a.downrate;   // hmc.py:188-191
}

template <int KMAX>
static hipError_t launch_poly(const PolyHmcArgs &a, bool fma, hipStream_t st)
{
    const dim3 grid((unsigned)((a.C + 63) / 64));
    if (fma) hmc_poly_small_kernel<KMAX, true><<<grid, 64, 0, st>>>(a);
    else     hmc_poly_small_kernel<KMAX, false><<<grid, 64, 0, st>>>(a);
    return hipGetLastError();
}

}  // namespace binf

using namespace binf;

extern "C" int32_t binf_hmc_sample_poly_f64(
    const double *q0, const double *p0, const double *u, double *q_out,
    uint8_t *accepted, int64_t *n_accepted, double *e_before, double *e_after,
    const double *xs, const double *ys, double precision,
    const double *precision_chain, const double *prior_means,
    const double *prior_vars, int32_t prior_first, const double *lp_pre,
    const double *lp_post, double timestep, double *dt_chain, int64_t C,
    int64_t K, int64_t N, int32_t nsteps, int32_t adapt, double uprate,
    double downrate, int32_t mode, void *stream)
{
    if (C < 0 || K < 1 || N < 0 || nsteps < 1)
        return fail(BINF_E_ARG, "hmc_sample_poly: need C>=0, K>=1, N>=0, nsteps>=1");
    const bool lane_per_chain = (mode & BINF_MODE_LANE_PER_CHAIN) != 0;
    mode &= ~BINF_MODE_LANE_PER_CHAIN;
    if (mode != BINF_MODE_EXACT && mode != BINF_MODE_FMA)
        return fail(BINF_E_ARG, "hmc_sample_poly: unknown mode %d", mode);
    if (K > 16 || N > 1024 || (N > 128 && pairwise_tree_height(N) > 3))
        return fail(BINF_E_UNSUPPORTED, "hmc_sample_poly: K=%lld > 16 or n_data=%lld > 1024 (or a pairwise tree deeper than 3) not covered by the fused kernels (use the per-step tier)", (long long)K, (long long)N);
    if (lane_per_chain && N > 128)
        return fail(BINF_E_UNSUPPORTED, "hmc_sample_poly: one lane per chain covers n_data <= 128, not %lld", (long long)N);
    if (C == 0) return 0;
    if (!q0 || !p0 || !u || !q_out || !accepted || (N > 0 && (!xs || !ys)))
        return fail(BINF_E_ARG, "hmc_sample_poly: null buffer");
    if ((prior_means == nullptr) != (prior_vars == nullptr))
        return fail(BINF_E_ARG, "hmc_sample_poly: prior_means and prior_vars go together");
    if (adapt && !dt_chain)
        return fail(BINF_E_ARG, "hmc_sample_poly: adaption needs dt_chain");
    if (C > 0x7fffffffLL * 64)
        return fail(BINF_E_UNSUPPORTED, "hmc_sample_poly: too many chains");
    const int64_t bytes = C * K * (int64_t)sizeof(double);
    const char *qo = (const char *)q_out, *pi = (const char *)p0, *qi = (const char *)q0;
    if ((qo != qi && qo < qi + bytes && qi < qo + bytes) || (qo < pi + bytes && pi < qo + bytes))
        return fail(BINF_E_ALIAS, "hmc_sample_poly: q_out overlaps q0/p0 (only q_out == q0 is allowed)");
    PolyHmcArgs a;
    a.q0 = q0; a.p0 = p0; a.u = u; a.q_out = q_out; a.accepted = accepted;
    a.n_accepted = n_accepted; a.e_before = e_before; a.e_after = e_after;
    a.xs = xs; a.ys = ys; a.tau_chain = precision_chain; a.tau = precision;
    a.prior_means = prior_means; a.prior_vars = prior_vars; a.lp_pre = lp_pre;
    a.lp_post = lp_post; a.dt_chain = dt_chain; a.timestep = timestep;
    a.uprate = uprate; a.downrate = downrate; a.C = C; a.K = (int32_t)K;
    a.N = (int32_t)N; a.nsteps = nsteps; a.prior_first = prior_first ? 1 : 0;
    a.adapt = adapt ? 1 : 0;
    const bool fma = (mode == BINF_MODE_FMA);
    hipStream_t st = (hipStream_t)stream;
    if (!lane_per_chain) return launch_poly_wave_from(a, fma, st);
    hipError_t e;
    if (K <= 4)      e = launch_poly<4>(a, fma, st);
    else if (K <= 8) e = launch_poly<8>(a, fma, st);
    else             e = launch_poly<16>(a, fma, st);
    if (e != hipSuccess) return hip_fail(e, "hmc_poly_small_kernel launch");
    return 0;
}

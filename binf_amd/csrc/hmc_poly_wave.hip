// Fused HMC transition on the example's polynomial posterior for MEDIUM data
// sets (128 < n_data <= 1024, K <= 16 coefficients): between the one-lane-per-chain
// kernel of hmc_poly.hip (n_data <= 128) and the MFMA gradient of the per-step tier
// (which needs ~3 L launches per transition and is launch-bound until
// chains x data is large).  Same entry point (binf_hmc_sample_poly_f64), same
// reference lines (binf/samplers/hmc.py:92-164 around posteriors.py:147-187,
// likelihoods.py:141-155, binf/example/likelihood.py:24-30,54-61, priors.py:49-54).
//
// Mapping: the DATA are spread over the G = 8 * 2^H lanes of a chain exactly as the
// Gaussian kernel spreads coordinates (H = height of numpy's pairwise tree for
// n_data; lane (leaf g, accumulator j) owns data points off_g + 8 t + j): the
// <= 16 (x, y) pairs of every lane slot sit in LDS ([t][slot], shared by all the
// chains of the workgroup, conflict-free), theta / p / the force are replicated in
// the registers of the chain's lanes.
//   chi^2   per-lane running sums in numpy's accumulator order + the xor-shuffle
//           tree of chain_sum_finish: bit-identical to np.sum((polyval - ys)**2),
//           so E_before / E_after carry the bits of the per-step tier;
//   force   per-lane FMA partial sums over the lane's data, then an xor-butterfly
//           over the chain's lanes (a + b == b + a, so every lane ends with the same
//           bits and the replicas never diverge); the order depends on n_data only,
//           not on the batch.  Held to the reference like every force here:
//           1e-10 of the sum-of-magnitudes scale (tests/poly_bounds.py).
#include "hmc_poly_args.hpp"

namespace binf {

// np.sum over K <= KMAX register values (every lane for itself): hmc_poly.hip
template <int KMAX, class F>
__device__ inline double np_sum_k(F f, int K)
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
__global__ void __launch_bounds__(256) hmc_poly_wave_kernel(const PolyHmcArgs a, const int32_t H)
{
    constexpr int TMAX = 16;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int K = a.K, N = a.N;
    const int lg = 3 + H;
    const int slot = lane & ((1 << lg) - 1);
    const int j = slot & 7;
    const Leaf Lf = pairwise_leaf(N, H, slot >> 3);
    const int n = Lf.len;
    const int T = (n >= 8) ? (n >> 3) : 0;
    const int rem = (n >= 8) ? (n & 7) : n;
    const int64_t raw = (wave << (6 - lg)) + (lane >> lg);
    const bool valid = raw < a.C;
    const int64_t c = valid ? raw : a.C - 1;

    // the data points of every lane slot, staged once per workgroup: sx[t][slot]
    __shared__ double sx[TMAX][64], sy[TMAX][64];
    for (int i = threadIdx.x; i < TMAX * 64; i += 256) {
        const int t = i >> 6, sl = i & 63;
        const Leaf L2 = pairwise_leaf(N, H, (sl & ((1 << lg) - 1)) >> 3);
        const int e = 8 * t + (sl & 7);
        const bool m = sl < (1 << lg) && e < L2.len;
        sx[t][sl] = m ? a.xs[L2.off + e] : 0.0;
        sy[t][sl] = m ? a.ys[L2.off + e] : 0.0;
    }
    __syncthreads();
    // a redundant path of a ragged tree recomputes its leaf for the energy tree but
    // must not count it twice in the force: its force weight is zero
    const double fcanon = Lf.canonical ? 1.0 : 0.0;
    double th[KMAX], p[KMAX], g[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        th[k] = (k < K) ? a.q0[c * K + k] : 0.0;
        p[k] = (k < K) ? a.p0[c * K + k] : 0.0;
    }
    const double tau = a.tau_chain ? a.tau_chain[c] : a.tau;
    const double dt = a.dt_chain ? a.dt_chain[c] : a.timestep;
    const double uu = a.u[c];
    const double logZ = (double)N * 0.5 * log(tau);               // likelihood.py:55
    const double lp_pre = a.lp_pre ? a.lp_pre[c] : 0.0;
    const double lp_post = a.lp_post ? a.lp_post[c] : 0.0;

    auto log_prob = [&]() {
        LaneSum s = {0.0, 0.0};
#pragma unroll 4
        for (int t = 0; t < TMAX; ++t) {
            // polyval: Horner, zero-padded above K-1 (exact no-ops)
            const double x = sx[t][slot];
            double v = th[KMAX - 1] + x * 0.0;
#pragma unroll
            for (int k = KMAX - 2; k >= 0; --k) v = th[k] + v * x;
            const double d = v - sy[t][slot];
            lane_sum_add<false>(s, d * d, t, T);
        }
        const double chi2 = 1.0 * chain_sum_finish<false, 0>(s, T, rem, lane, H, Lf.depth);
        const double lik = -0.5 * chi2 * tau + logZ;              // likelihood.py:56-57
        double pri = 0.0;
        if (a.prior_means) {
            auto term = [&](int k) {
                const double d = th[k] - ((k < K) ? a.prior_means[k] : 0.0);
                return d * d / ((k < K) ? a.prior_vars[k] : 1.0);  // priors.py:52-54
            };
            pri = -0.5 * np_sum_k<KMAX>(term, K);
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
        return 0.5 * np_sum_k<KMAX>(sq, K);
    };
    // force = tau * sum_n (polyval(x_n) - y_n) x_n^k             likelihoods.py:148-155
    auto force = [&]() {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) g[k] = 0.0;
#pragma unroll 4
        for (int t = 0; t < TMAX; ++t) {
            const double x = sx[t][slot];
            double v = th[KMAX - 1];
#pragma unroll
            for (int k = KMAX - 2; k >= 0; --k) v = __builtin_fma(v, x, th[k]);
            // rows past the lane's data hold x = y = 0 with theta_0 as "residual": masked
            const double r = (8 * t + j < n) ? (v - sy[t][slot]) * tau * fcanon : 0.0;
            double pw = 1.0;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                g[k] = __builtin_fma(pw, r, g[k]);
                pw = pw * x;
            }
        }
        // all-reduce over the chain's lanes; padded coefficients stay exactly zero
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            double s = g[k];
            for (int m = 1; m < (1 << lg); m <<= 1) s = s + shfl_xor_f64(s, m);
            g[k] = (k < K) ? s : 0.0;
        }
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
    if (!valid || slot != 0) return;
    // q_out may be q0 itself: a rejected chain then keeps its state untouched
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) {
            const double old = a.q0[c * K + k];
            a.q_out[c * K + k] = acc ? th[k] : old;
        }
    a.accepted[c] = acc ? 1 : 0;
    if (a.n_accepted && acc) a.n_accepted[c] += 1;
    if (a.e_before) a.e_before[c] = e_before;
    if (a.e_after) a.e_after[c] = e_after;
    if (a.adapt) a.dt_chain[c] = acc ? dt * a.uprate : dt * a.downrate;   // hmc.py:188-191
}

template <int KMAX>
static hipError_t launch_poly_wave_k(const PolyHmcArgs &a, int32_t H, bool fma, hipStream_t st)
{
    const int64_t chains_per_wave = 64 >> (3 + H);
    const int64_t waves = (a.C + chains_per_wave - 1) / chains_per_wave;
    const dim3 grid((unsigned)((waves + 3) / 4));
    if (fma) hmc_poly_wave_kernel<KMAX, true><<<grid, 256, 0, st>>>(a, H);
    else     hmc_poly_wave_kernel<KMAX, false><<<grid, 256, 0, st>>>(a, H);
    return hipGetLastError();
}

int32_t launch_poly_wave_from(const PolyHmcArgs &a, bool fma, hipStream_t st)
{
    const int32_t H = pairwise_tree_height(a.N);
    hipError_t e;
    if (a.K <= 4)      e = launch_poly_wave_k<4>(a, H, fma, st);
    else if (a.K <= 8) e = launch_poly_wave_k<8>(a, H, fma, st);
    else               e = launch_poly_wave_k<16>(a, H, fma, st);
    if (e != hipSuccess) return hip_fail(e, "hmc_poly_wave_kernel launch");
    return 0;
}

}  // namespace binf

// The example's polynomial posterior, one chain per lane GROUP, any number of
// Gibbs sweeps per launch.  One kernel template serves
//
//   binf_hmc_sample_poly_f64   one HMCSampler.sample() (hmc_poly_wave.hip):
//                              GIBBS = false, n = 1, draws and the constant
//                              log-prob terms supplied by the caller;
//   binf_gibbs_poly_sample_n_f64   n sweeps of the example's Gibbs loop
//                              (gibbs_poly.hip), example_script.py:33-34 around
//                              binf/samplers/gibbs.py:146-149:
//                                coefficients: HMCSampler.sample (hmc.py:136-164)
//                                              or RWMCSampler.sample
//                                              (binf/example/samplers.py:78-92)
//                                precision:    GammaSampler.sample
//                                              (binf/example/samplers.py:27-51)
//                              with the state of a chain in registers between the
//                              sweeps and, if no draws are supplied, the draws of
//                              samplers/rng.py:DeviceRNG generated in place.
//
// so n sweeps in one launch run the very instructions of n single launches:
// bit-identical by construction (tests/test_gpu_gibbs_n.py).
//
// Mapping: the DATA are spread over the G = 8 * 2^H lanes of a chain exactly as the
// Gaussian kernel spreads coordinates (H = height of numpy's pairwise tree for
// n_data, H = 0 up to 128 points; lane (leaf g, accumulator j) owns data points
// off_g + 8 t + j): the (x, y) pairs of every lane slot sit in LDS ([t][slot],
// shared by all the chains of the workgroup, conflict-free), theta / p / the force
// are replicated in the registers of the chain's lanes.
//   chi^2   per-lane running sums in numpy's accumulator order + the xor-shuffle
//           tree of chain_sum_finish: bit-identical to np.sum((polyval - ys)**2),
//           so E_before / E_after carry the bits of the per-step tier;
//   force   per-lane FMA partial sums over the lane's data, then an xor-butterfly
//           over the chain's lanes (a + b == b + a, so every lane ends with the same
//           bits and the replicas never diverge); the order depends on n_data only,
//           not on the batch.  Held to the reference like every force here:
//           1e-10 of the sum-of-magnitudes scale (tests/poly_bounds.py).
// gfx950, wave64.
#pragma once
#include "gauss_common.hpp"
#include "philox_draws.hpp"
#include <type_traits>

namespace binf {

struct PolyChainArgs {
    const double *theta0;      // [C x K]
    const double *tau0;        // [C] or null (then `tau`)
    double tau;
    double *theta_out;         // [C x K]; may be theta0
    double *tau_out;           // [C] (GIBBS) or null
    double *rec_theta;         // [n / thin x C x K] or null
    double *rec_tau;           // [n / thin x C] or null
    uint8_t *accepted;         // [n x C] or null
    int64_t *n_accepted;       // [C] or null
    double *e_before;          // [n x C] or null (HMC move)
    double *e_after;           // [n x C] or null
    const double *xs;          // [N]
    const double *ys;          // [N]
    const double *prior_means; // [K] or null: Gaussian prior on theta (energy only)
    const double *prior_vars;  // [K]
    const double *lp_pre;      // !GIBBS: [C] or null, theta-independent terms added first
    const double *lp_post;     // !GIBBS: ... added last
    const double *p0;          // [n x C x K] momenta (HMC) / proposal steps (RWMC); null: generated
    const double *u;           // [n x C] acceptance draws; null: generated
    const double *g;           // [n x C] Gamma(shape, 1) variates; null: generated
    double *dt_chain;          // [C] or null
    double timestep;
    double uprate;
    double downrate;
    double stepsize;           // RWMC half-width
    double gp_shape_m1;        // GIBBS: the GammaPrior term of the coefficient conditional,
    double gp_rate;            //        (shape - 1) log tau - tau rate (priors.py:23-25)
    double g_shape;            // GIBBS: shape of the conjugate draw (samplers.py:27-32)
    double g_rate;             //        prior rate added to 0.5 chi^2 (samplers.py:34-41)
    int64_t C;
    int64_t chain_offset;      // global index of chain 0 of this launch (generated draws)
    uint64_t seed_m, off_m, stride_m;   // momentum / proposal stream: sweep i at off_m + i stride_m
    uint64_t seed_u, off_u, stride_u;   // acceptance draws
    uint64_t seed_g, off_g, stride_g;   // gamma variates
    int32_t K;
    int32_t N;
    int32_t H;                 // pairwise tree height of N
    int32_t tcount;            // rounds of 8 data points per leaf: ceil(longest leaf / 8)
    int32_t nsteps;
    int32_t n;                 // sweeps (GIBBS) / 1
    int32_t thin;
    int32_t n_adapt;           // the first n_adapt HMC transitions adapt the timestep
    int32_t prior_first;       // the Gaussian prior term precedes the likelihood term
    int32_t gp_where;          // GIBBS: 0 no GammaPrior term, 1 before the theta terms, 2 after
    int32_t zig;               // generated momenta: 1 ziggurat, 0 Box-Muller (rng.hip streams)
    int32_t keep_tau;          // GIBBS: no precision draw (n moves under a fixed precision)
};

constexpr int POLY_MOVE_HMC = 0;
constexpr int POLY_MOVE_RWMC = 1;

// np.sum over K <= KMAX register values (every lane for itself)
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

// both outputs of block i of the ziggurat normal stream (rng.hip): elements 2i, 2i + 1
__device__ inline void zig_normal_pair(int64_t i, uint64_t seed, uint64_t offset, const double *zx,
                                       const double *zr, double &a, double &b)
{
    const Philox4 r = zig_block(i, seed, offset, 0);
    int layer;
    double u;
    zig_split(r.v[0], r.v[1], layer, u);
    a = (fabs(u) < zr[layer]) ? u * zx[layer] : zig_slow(r.v[0], r.v[1], zx, zr, i, seed, offset, 0);
    zig_split(r.v[2], r.v[3], layer, u);
    b = (fabs(u) < zr[layer]) ? u * zx[layer] : zig_slow(r.v[2], r.v[3], zx, zr, i, seed, offset, 1);
}

template <int KMAX, bool FMA, bool GIBBS, int MOVE>
__global__ void __launch_bounds__(256) poly_chain_kernel(const PolyChainArgs a)
{
    constexpr int TMAX = 16;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int K = a.K, N = a.N, H = a.H;
    const int lg = 3 + H;
    const int slot = lane & ((1 << lg) - 1);
    const int chainbase = lane - slot;
    const int j = slot & 7;
    const Leaf Lf = pairwise_leaf(N, H, slot >> 3);
    const int n = Lf.len;
    const int T = (n >= 8) ? (n >> 3) : 0;
    const int rem = (n >= 8) ? (n & 7) : n;
    const int TC = a.tcount;
    const int64_t raw = (wave << (6 - lg)) + (lane >> lg);
    const bool valid = raw < a.C;
    const int64_t c = valid ? raw : a.C - 1;

    // the data points of every lane slot, staged once per workgroup: sx[t][slot]
    __shared__ double sx[TMAX][64], sy[TMAX][64];
    __shared__ double zx[GIBBS ? ZIG_C + 1 : 1], zr[GIBBS ? ZIG_C : 1];
    for (int i = threadIdx.x; i < TMAX * 64; i += 256) {
        const int t = i >> 6, sl = i & 63;
        const Leaf L2 = pairwise_leaf(N, H, (sl & ((1 << lg) - 1)) >> 3);
        const int e = 8 * t + (sl & 7);
        const bool m = sl < (1 << lg) && e < L2.len;
        sx[t][sl] = m ? a.xs[L2.off + e] : 0.0;
        sy[t][sl] = m ? a.ys[L2.off + e] : 0.0;
    }
    if (GIBBS && MOVE == POLY_MOVE_HMC && !a.p0 && a.zig) {
        for (int k = threadIdx.x; k <= ZIG_C; k += 256) zx[k] = ZIG_X[k];
        for (int k = threadIdx.x; k < ZIG_C; k += 256) zr[k] = ZIG_RATIO[k];
    }
    __syncthreads();
    // a redundant path of a ragged tree recomputes its leaf for the energy tree but
    // must not count it twice in the force: its force weight is zero
    const double fcanon = Lf.canonical ? 1.0 : 0.0;
    double th[KMAX], p[KMAX], g[KMAX], old[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) th[k] = (k < K) ? a.theta0[c * K + k] : 0.0;
    double tau = a.tau0 ? a.tau0[c] : a.tau;
    double dt = a.dt_chain ? a.dt_chain[c] : a.timestep;

    // np.sum((polyval(xs, theta) - ys)**2): Horner, zero-padded above K-1 (exact no-ops)
    // A lane's data points are visited in rounds t = 0, 1, ... (point 8 t + j of its leaf).
    // With one wave per SIMD or less -- a few thousand chains -- every dependent FP64
    // operation costs its full latency, so the rounds run FOUR AT A TIME with their Horner
    // chains interleaved; rounds 0..3 live in registers (20 data points are 3 rounds),
    // later ones come from LDS.  Rounds past a lane's data hold x = y = 0 and contribute
    // exact zeros / are masked, so the grouping changes no bit.
    double xr[4], yr[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        xr[t] = sx[t][slot];
        yr[t] = sy[t][slot];
    }
    const int groups = (TC + 3) >> 2;
    auto chi2_of = [&]() {
        LaneSum s = {0.0, 0.0};
        auto four = [&](const double (&x)[4], const double (&y)[4], int t0) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = th[KMAX - 1] + x[u] * 0.0;
#pragma unroll
            for (int k = KMAX - 2; k >= 0; --k) {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = th[k] + v[u] * x[u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double d = v[u] - y[u];
                lane_sum_add<false>(s, d * d, t0 + u, T);
            }
        };
        four(xr, yr, 0);
        for (int gi = 1; gi < groups; ++gi) {
            double x[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x[u] = sx[4 * gi + u][slot];
                y[u] = sy[4 * gi + u][slot];
            }
            four(x, y, 4 * gi);
        }
        return 1.0 * chain_sum_finish<false, 0>(s, T, rem, lane, H, Lf.depth);
    };
    // log posterior of theta given tau: the component terms added one after the other in
    // the Posterior's order (posteriors.py:147-151, sorted component names)
    auto log_prob = [&](double chi2, double logZ, bool have_pre, double cpre, bool have_post,
                        double cpost) {
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
        if (have_pre) add(cpre);
        if (a.prior_means && a.prior_first) add(pri);
        add(lik);
        if (a.prior_means && !a.prior_first) add(pri);
        if (have_post) add(cpost);
        return total;
    };
    auto kinetic = [&]() {                                        // hmc.py:148,150
        auto sq = [&](int k) { return p[k] * p[k]; };
        return 0.5 * np_sum_k<KMAX>(sq, K);
    };
    // force = tau * sum_n (polyval(x_n) - y_n) x_n^k             likelihoods.py:148-155
    // R rounds at a time, their Horner chains interleaved.  m[u] = tau for a data point the
    // lane owns and counts, 0 for a round past its data (x = y = 0 there) or on a redundant
    // path of a ragged tree: (v - y) * m is the residual times tau, or an exact zero.
    auto force_rounds = [&](auto rc, const double *x, const double *y, const double *m) {
        constexpr int R = decltype(rc)::value;
        double v[R], r[R], pw[R];
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = th[KMAX - 1];
#pragma unroll
        for (int k = KMAX - 2; k >= 0; --k) {
#pragma unroll
            for (int u = 0; u < R; ++u) v[u] = __builtin_fma(v[u], x[u], th[k]);
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            r[u] = (v[u] - y[u]) * m[u];
            pw[u] = 1.0;
        }
        // every g[k] receives its terms in the order t = 0, 1, 2, ...
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                g[k] = __builtin_fma(pw[u], r[u], g[k]);
                if (k + 1 < KMAX) pw[u] = pw[u] * x[u];
            }
        }
    };
    double mr[4];                      // the multipliers of rounds 0..3 under the current tau
    auto force = [&]() {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) g[k] = 0.0;
        if (TC <= 2) {
            force_rounds(std::integral_constant<int, 2>(), xr, yr, mr);
        } else if (TC == 3) {
            force_rounds(std::integral_constant<int, 3>(), xr, yr, mr);
        } else {
            force_rounds(std::integral_constant<int, 4>(), xr, yr, mr);
            for (int gi = 1; gi < groups; ++gi) {
                double x[4], y[4], m[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    x[u] = sx[4 * gi + u][slot];
                    y[u] = sy[4 * gi + u][slot];
                    m[u] = (8 * (4 * gi + u) + j < n) ? tau * fcanon : 0.0;
                }
                force_rounds(std::integral_constant<int, 4>(), x, y, m);
            }
        }
        // all-reduce over the chain's lanes, level by level for all coefficients at once
        // (independent chains for the pipeline); padded coefficients stay exactly zero
#pragma unroll
        for (int k = 0; k < KMAX; ++k) g[k] = g[k] + xor1_f64(g[k]);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) g[k] = g[k] + xor2_f64(g[k]);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) g[k] = g[k] + other_quad_f64(g[k]);
        if (lg > 3) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) g[k] = g[k] + xor8_f64(g[k]);
            if (lg > 4) {
#pragma unroll
                for (int k = 0; k < KMAX; ++k) g[k] = g[k] + xor16_f64(g[k], lane);
            }
            if (lg > 5) {
#pragma unroll
                for (int k = 0; k < KMAX; ++k) g[k] = g[k] + xor32_f64(g[k], lane);
            }
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) g[k] = (k < K) ? g[k] : 0.0;
    };
    // K draws of a chain from a paired Philox stream: lane (slot & 7) of the chain computes
    // block (e0 >> 1) + (slot & 7) -- global elements 2b, 2b + 1 -- and the chain's lanes
    // pick element e0 + k from the lane that holds it
    auto gather_pairs = [&](int64_t e0, double va, double vb) {
        const int odd = (int)(e0 & 1);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int s = k + odd;
            const double sel = (s & 1) ? vb : va;
            const double v = shfl_f64(sel, chainbase + ((s >> 1) & 7));
            p[k] = (k < K) ? v : 0.0;
        }
    };

    double chi2 = chi2_of();            // of the current state; carried from sweep to sweep
    int32_t nacc = 0;
    const int64_t gc = a.chain_offset + c;      // global chain index (generated draws)
    const int nsweeps = GIBBS ? a.n : 1;     // a constant trip count keeps the single launch lean
    for (int i = 0; i < nsweeps; ++i) {
        const double logZ = (double)N * 0.5 * log(tau);           // likelihood.py:55
        bool have_pre, have_post;
        double cpre = 0.0, cpost = 0.0;
        if (GIBBS) {
            const double gp = a.gp_shape_m1 * log(tau) - tau * a.gp_rate;   // priors.py:23-25
            have_pre = a.gp_where == 1;
            have_post = a.gp_where == 2;
            cpre = cpost = gp;
        } else {
            have_pre = a.lp_pre != nullptr;
            have_post = a.lp_post != nullptr;
            cpre = have_pre ? a.lp_pre[c] : 0.0;
            cpost = have_post ? a.lp_post[c] : 0.0;
        }
        // ---- the draws of this sweep ----------------------------------------------------
        const int64_t ic = (int64_t)i * a.C + c;
        if (a.p0) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) p[k] = (k < K) ? a.p0[ic * K + k] : 0.0;
        } else if (GIBBS) {
            const int64_t e0 = gc * K;
            const int64_t b = (e0 >> 1) + (slot & 7);
            const uint64_t off = a.off_m + (uint64_t)i * a.stride_m;
            double va, vb;
            if (MOVE == POLY_MOVE_HMC) {                          // hmc.py:146
                if (a.zig) zig_normal_pair(b, a.seed_m, off, zx, zr, va, vb);
                else       normals2(b, a.seed_m, off, va, vb);
            } else {                                              // samplers.py:80-81
                const double low = -a.stepsize;
                const double scale = a.stepsize - low;
                uniforms2(b, a.seed_m, off, va, vb);
                va = low + scale * va;
                vb = low + scale * vb;
            }
            gather_pairs(e0, va, vb);
        }
        double uu;
        if (a.u) uu = a.u[ic];
        else uu = GIBBS ? uniform_elem(gc, a.seed_u, a.off_u + (uint64_t)i * a.stride_u) : 0.0;

        // ---- the move of the coefficients ------------------------------------------------
        bool acc;
        double chi2_new, e_before = 0.0, e_after = 0.0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) old[k] = th[k];
        if (MOVE == POLY_MOVE_HMC) {
            e_before = -log_prob(chi2, logZ, have_pre, cpre, have_post, cpost) + kinetic();  // hmc.py:148
            const double hdt = 0.5 * dt;
#pragma unroll
            for (int u = 0; u < 4; ++u) mr[u] = (8 * u + j < n) ? tau * fcanon : 0.0;
            // hmc.py:116-123 as ONE loop around the force: half kick, (nsteps - 1) x [drift,
            // kick], drift, half kick
            for (int l = 0; l <= a.nsteps; ++l) {
                force();
                const double kdt = (l == 0 || l == a.nsteps) ? hdt : dt;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) p[k] = kick<FMA>(p[k], kdt, g[k]);
                if (l < a.nsteps) {
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) th[k] = drift<FMA>(th[k], p[k], dt);
                }
            }
            chi2_new = chi2_of();
            e_after = -log_prob(chi2_new, logZ, have_pre, cpre, have_post, cpost) + kinetic();  // hmc.py:150
            double x = -(e_after - e_before);                            // hmc.py:151
            x = (x < -308.0) ? -308.0 : x;
            x = (x > 709.0) ? 709.0 : x;
            acc = uu < exp_clipped_range(x);
        } else {
            // E_old = -log_prob(state), proposal = state + change, E_new (samplers.py:78-84);
            // -(E_new - E_old) == lp_new - lp_old bit for bit (rwmc.hip)
            const double lp_old = log_prob(chi2, logZ, have_pre, cpre, have_post, cpost);
#pragma unroll
            for (int k = 0; k < KMAX; ++k) th[k] = (k < K) ? th[k] + p[k] : 0.0;
            chi2_new = chi2_of();
            const double lp_new = log_prob(chi2_new, logZ, have_pre, cpre, have_post, cpost);
            acc = uu < np_exp(lp_new - lp_old);                          // samplers.py:86
        }
        if (GIBBS) {
            // the chain's lanes hold replicas: slot 0 decides (it is the lane whose results
            // a single launch writes out)
            acc = __shfl((int)acc, chainbase, 64) != 0;
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) th[k] = acc ? th[k] : old[k];
        chi2 = acc ? chi2_new : chi2;
        nacc += acc ? 1 : 0;
        if (MOVE == POLY_MOVE_HMC && i < a.n_adapt)
            dt = acc ? dt * a.uprate : dt * a.downrate;                   // hmc.py:188-191

        // ---- the conjugate draw of the precision (samplers.py:27-51) -----------------------
        if (GIBBS && !a.keep_tau) {
            // likelihood.log_prob(coefficients, precision=1.0): the chi^2 epilogue of
            // rowsum.hpp at tau = 1
            const double lp1 = -0.5 * chi2 * 1.0 + (double)N * 0.5 * log(1.0);
            const double rate = -lp1 + a.g_rate;
            const double gv = a.g ? a.g[ic]
                                  : gamma_elem<false>(gc, a.g_shape, a.seed_g,      // shape >= 1
                                                      a.off_g + (uint64_t)i * a.stride_g);
            tau = gv / rate;
            tau = shfl_f64(tau, chainbase);
        }
        if (valid && slot == 0) {
            if (a.accepted) a.accepted[ic] = acc ? 1 : 0;
            if (a.e_before) a.e_before[ic] = e_before;
            if (a.e_after) a.e_after[ic] = e_after;
            if (GIBBS && (i + 1) % a.thin == 0) {
                const int64_t r = (int64_t)((i + 1) / a.thin - 1) * a.C + c;
                if (a.rec_theta) {
#pragma unroll
                    for (int k = 0; k < KMAX; ++k)
                        if (k < K) a.rec_theta[r * K + k] = th[k];
                }
                if (a.rec_tau) a.rec_tau[r] = tau;
            }
        }
    }
    if (!valid || slot != 0) return;
    // theta_out may be theta0 itself: a chain then keeps or replaces its own row
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) a.theta_out[c * K + k] = th[k];
    if (GIBBS && a.tau_out) a.tau_out[c] = tau;
    if (a.n_accepted && nacc) a.n_accepted[c] += nacc;
    if (a.n_adapt > 0 && a.dt_chain) a.dt_chain[c] = dt;
}

}  // namespace binf

// Polynomial forward model + Gaussian error model (BASELINE configs C1/C3/C4).
// gfx950 (MI355X), wave64.
//
// Reference lines replaced:
//   ForwardModel._evaluate / _evaluate_jacobi_matrix  binf/example/likelihood.py:24-30
//   GaussianErrorModel log_prob / gradient            binf/example/likelihood.py:54-61
//   Likelihood._evaluate_log_prob / _evaluate_gradient binf/pdf/likelihoods.py:141-155
//   GammaSampler rate / draw                          binf/example/samplers.py:34-51
//
// log-prob: Horner evaluation in numpy.polynomial.polyval's order, squared
// residuals summed in np.sum's order -> bit-identical chi^2.
// gradient: G = (Theta.A - y) tau . A^T with the [K x N] design matrix A, as
// two chained f64 MFMA products per 16x16 tile; the [C x N] mock data never
// leaves registers.  BLAS summation order is not reproducible, so this path
// is held to the reference by tolerance, not bitwise.
#include "rowsum.hpp"

namespace binf {

typedef double v4d __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// Horner, numpy.polynomial.polynomial.polyval order:
//   c0 = c[-1] + x*0;  for i in 2..K: c0 = c[-i] + c0*x
// Coefficients above K-1 are zero-padded: 0 + 0*x steps are exact no-ops for
// finite x, so one KMAX instantiation serves every K <= KMAX bit-exactly.
// ---------------------------------------------------------------------------
template <int KMAX>
struct Horner {
    double c[KMAX];
    __device__ inline void load(const double *theta, int K)
    {
#pragma unroll
        for (int i = 0; i < KMAX; ++i) c[i] = (i < K) ? theta[i] : 0.0;
    }
    __device__ inline double operator()(double x) const
    {
        double v = c[KMAX - 1] + x * 0.0;
#pragma unroll
        for (int i = KMAX - 2; i >= 0; --i) v = c[i] + v * x;
        return v;
    }
};

struct PolyArgs {
    const double *theta;   // [C x K]
    const double *xs;      // [N]
    const double *ys;      // [N]
    int32_t K;
};

template <int KMAX>
struct ResidSq {
    Horner<KMAX> h;
    const double *xs;
    const double *ys;
    __device__ inline double operator()(int i) const
    {
        const double d = h(xs[i]) - ys[i];
        return d * d;
    }
};

template <int KMAX>
struct ResidSqMake {
    __device__ static inline ResidSq<KMAX> make(const PolyArgs &a, int64_t row)
    {
        ResidSq<KMAX> f;
        f.h.load(a.theta + row * a.K, a.K);
        f.xs = a.xs;
        f.ys = a.ys;
        return f;
    }
};

// mock[c, n] = polyval(xs[n], theta[c, :])
template <int KMAX>
__global__ void __launch_bounds__(256)
poly_forward_kernel(const PolyArgs a, double *out, int64_t C, int64_t N)
{
    const int64_t c = blockIdx.y;
    Horner<KMAX> h;
    h.load(a.theta + c * a.K, a.K);
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N;
         n += (int64_t)gridDim.x * 256)
        out[c * N + n] = h(a.xs[n]);
}

// out[c, n] = (mock[c, n] - ys[n]) * tau_c          likelihood.py:59-61
__global__ void __launch_bounds__(256)
gauss_err_grad_kernel(const double *mock, const double *ys, double tau,
                      const double *tau_chain, double *out, int64_t N)
{
    const int64_t c = blockIdx.y;
    const double t = tau_chain ? tau_chain[c] : tau;
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N;
         n += (int64_t)gridDim.x * 256)
        out[c * N + n] = (mock[c * N + n] - ys[n]) * t;
}

// tau[c] = g[c] / (-lp1[c] + prior_rate)                     samplers.py:34-51
__global__ void gamma_update_kernel(const double *g, const double *lp1,
                                    double prior_rate, double *out, int64_t C)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double rate = -lp1[c] + prior_rate;
    out[c] = g[c] / rate;
}

// GammaPrior._evaluate_log_prob: (shape - 1) * log(tau) - tau * rate      priors.py:10-25
__global__ void gamma_logp_kernel(const double *tau, double shape_m1, double rate, double *out,
                                  int64_t C)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double t = tau[c];
    out[c] = shape_m1 * log(t) - t * rate;
}

// ---------------------------------------------------------------------------
// gradient: two chained f64 MFMA products per (16 data points x 16 chains)
// ---------------------------------------------------------------------------
struct GradArgs {
    const double *theta;     // [C x K]
    const double *A;         // [K x N] design matrix, A[i][n] = xs[n]**i
    const double *ys;        // [N]
    const double *tau_chain; // [C] or null
    double tau;
    double *part;            // [NS x C x K] partial sums (or the output if NS==1)
    int64_t C;
    int32_t K;
    int32_t N;
    int32_t tiles_per_split; // data tiles (of 16) per blockIdx.y
};

constexpr int LDA = 17;      // padded row length of the staged A tile (doubles)

// KS forward k-steps (coefficients 0 .. 4*KS-1), RT backward row tiles
// (coefficients 0 .. 16*RT-1) on the MFMA pipe; KV trailing coefficients
// (indices KB .. KB+KV-1, KB = 4*KS = 16*RT when KV > 0) on the VALU, which
// runs beside the MFMA pipe: for K = 33 that is 16 MFMAs + 8 FMAs per tile
// instead of 21 MFMAs.
// CT = 16-chain tiles per wave: a wave owns CT * 16 chains, a workgroup
// 64 * CT.  The A-operand LDS reads are shared by the wave's tiles and the
// tiles' accumulator chains are independent, so the MFMA pipe sees CT times
// more work per barrier.
template <int KS, int RT, int KV, int CT>
__global__ void __launch_bounds__(256) poly_grad_mfma_kernel(const GradArgs a)
{
    constexpr int KB = 4 * KS;                       // first VALU coefficient
    constexpr int ROWS0 = (4 * KS > 16 * RT) ? 4 * KS : 16 * RT;
    constexpr int ROWS = ROWS0 + KV;
    constexpr int PASSES = (ROWS + 15) / 16;
    static_assert(KV == 0 || 4 * KS == 16 * RT, "VALU tail needs 4*KS == 16*RT");
    __shared__ double sA[2][ROWS][LDA];
    __shared__ double sY[2][16];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lc = lane & 15;
    const int lk = lane >> 4;
    const int K = a.K, N = a.N;

    int64_t chain[CT];
    bool cvalid[CT];
    double th[CT][KS];
    double thv[CT][KV > 0 ? KV : 1], gv[CT][KV > 0 ? KV : 1];
    double tau[CT];
    v4d G[CT][RT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        chain[c] = ((int64_t)blockIdx.x * 4 + wave) * (16 * CT) + 16 * c + lc;
        cvalid[c] = chain[c] < a.C;
        // forward B operands: theta[chain][4s + lk]
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 4 * s + lk;
            th[c][s] = (cvalid[c] && k < K) ? a.theta[chain[c] * K + k] : 0.0;
        }
#pragma unroll
        for (int v = 0; v < KV; ++v) {
            thv[c][v] = (cvalid[c] && KB + v < K) ? a.theta[chain[c] * K + KB + v] : 0.0;
            gv[c][v] = 0.0;
        }
        tau[c] = cvalid[c] ? (a.tau_chain ? a.tau_chain[chain[c]] : a.tau) : 0.0;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) G[c][rt] = (v4d){0.0, 0.0, 0.0, 0.0};
    }

    const int t0 = blockIdx.y * a.tiles_per_split;
    int t1 = t0 + a.tiles_per_split;
    const int ntiles = (N + 15) / 16;
    if (t1 > ntiles) t1 = ntiles;

    // staging: thread (srow, scol) moves A[16*pass + srow][n0 + scol].  The
    // loads are UNCONDITIONAL (indices clamped, invalid entries zeroed when
    // they are written to LDS): a load under a per-lane branch makes the
    // compiler wait for it at the join, i.e. before the tile's first MFMA,
    // and the prefetch would hide nothing.
    const int srow = tid >> 4, scol = tid & 15;
    double pre[PASSES];
    double prey = 0.0;
    auto fetch = [&](int t) {
        const int n = t * 16 + scol;
        const int nc = n < N ? n : N - 1;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int k = 16 * ps + srow;
            pre[ps] = a.A[(int64_t)(k < K ? k : K - 1) * N + nc];
        }
        prey = a.ys[nc];
    };
    auto stash = [&](int buf, int t) {
        const bool nv = t * 16 + scol < N;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int k = 16 * ps + srow;
            if (k < ROWS) sA[buf][k][scol] = (k < K && nv) ? pre[ps] : 0.0;
        }
        if (tid < 16) sY[buf][tid] = nv ? prey : 0.0;
    };

    if (t0 < t1) {
        fetch(t0);
        stash(0, t0);
    }
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        fetch(t + 1 < t1 ? t + 1 : t);         // in flight during the MFMAs
        __builtin_amdgcn_sched_barrier(0);     // ... so keep the loads up here
        // forward: M^T[n][c] = sum_k A[k][n] theta[c][k]
        v4d acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const double av = sA[buf][4 * s + lk][lc];
#pragma unroll
            for (int c = 0; c < CT; ++c)
                acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, th[c][s], acc[c], 0, 0, 0);
        }
        // trailing coefficients on the VALU: lane holds (n = lk + 4r, c = lc)
#pragma unroll
        for (int v = 0; v < KV; ++v)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double av = sA[buf][KB + v][lk + 4 * r];
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    acc[c][r] = __builtin_fma(thv[c][v], av, acc[c][r]);
            }
        // error-model gradient in place: r[n][c] = (mock - y[n]) * tau_c
        double rr[CT][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nl = lk + 4 * r;
            const double yv = sY[buf][nl];
            const bool nv = t * 16 + nl < N;
#pragma unroll
            for (int c = 0; c < CT; ++c)
                rr[c][r] = nv ? (acc[c][r] - yv) * tau[c] : 0.0;
        }
        // backward: G^T[i][c] += sum_n A[i][n] r[n][c]; the D-layout register r
        // of the forward product is exactly the B operand of k-step r.
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double av = sA[buf][16 * rt + lc][4 * s + lk];
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    G[c][rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, rr[c][s], G[c][rt], 0, 0, 0);
            }
#pragma unroll
        for (int v = 0; v < KV; ++v)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double av = sA[buf][KB + v][lk + 4 * r];
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    gv[c][v] = __builtin_fma(av, rr[c][r], gv[c][v]);
            }
        if (t + 1 < t1) stash(buf ^ 1, t + 1);
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        // the VALU rows: sum the four lk partials of each chain
#pragma unroll
        for (int v = 0; v < KV; ++v) {
            gv[c][v] = gv[c][v] + shfl_xor_f64(gv[c][v], 16);
            gv[c][v] = gv[c][v] + shfl_xor_f64(gv[c][v], 32);
        }
        if (cvalid[c]) {
            double *dst = a.part + ((int64_t)blockIdx.y * a.C + chain[c]) * K;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * rt + lk + 4 * r;
                    if (i < K) dst[i] = G[c][rt][r];
                }
            if (lk == 0) {
#pragma unroll
                for (int v = 0; v < KV; ++v)
                    if (KB + v < K) dst[KB + v] = gv[c][v];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// The same product for data sets made of WHOLE tiles (N % 16 == 0, K * N below
// 2^28 entries -- BASELINE C3 / C4: N = 16384): what the loop above spends beside
// its MFMAs is taken out of it.
//
// Why it matters (scripts/mfma64_duty.hip, profiles/r04_b_mfma64_duty.jsonl): on
// gfx950 NOTHING issues beside a v_mfma_f64_16x16x4_f64 for free.  A bare loop of
// them sustains 70.8 TFLOP/s (71 cycles per MFMA and SIMD at 2.39 GHz, 0.90 of the
// 78.6 datasheet figure); every VALU instruction a wave adds -- FP64, FP32, integer,
// v_mov alike -- costs the SIMD's matrix pipe another 3.4 - 5.9 cycles, whereas idle
// cycles and ds_read are hidden.  The general kernel issues ~100 VALU instructions
// per 32 MFMAs (address arithmetic for the prefetch and both LDS buffers, AGPR
// read-back of the forward accumulators, tail selects): 56 - 59 TFLOP/s.  Here:
//   * the tile loop is unrolled over the two LDS buffers, so every LDS address is a
//     per-thread base plus an immediate offset (no VALU);
//   * the prefetch addresses are a uniform tile pointer (SGPRs) plus a per-thread
//     32-bit offset computed once (no VALU, no clamp: every tile is whole);
//   * no validity selects (there is no partial tile; rows of A beyond K are copies of
//     row K - 1 that only ever meet zero coefficients or unstored output rows);
//   * the file is compiled with -amdgpu-mfma-vgpr-form: accumulators live in VGPRs,
//     the 16 v_accvgpr_read per tile between the two products are gone.
// The MFMA sequence, the split of the data range and the order of every sum are
// those of the general kernel: the two return the same bits.
// ---------------------------------------------------------------------------
template <int B> struct BufC { static constexpr int value = B; };

// Two workgroups (8 waves) per CU are what the launch shape asks for (grad_splits: 2 waves per
// SIMD keep the matrix pipe fed, a third buys nothing -- 66.9 vs 66.4 TFLOP/s at C3); the
// register allocator gets the 256 VGPRs that leaves it, so no instantiation spills.
template <int KS, int RT, int KV, int CT>
__global__ void __launch_bounds__(256, 2) poly_grad_mfma_full_kernel(const GradArgs a)
{
    constexpr int KB = 4 * KS;
    constexpr int ROWS0 = (4 * KS > 16 * RT) ? 4 * KS : 16 * RT;
    constexpr int ROWS = ROWS0 + KV;
    constexpr int PASSES = (ROWS + 15) / 16;
    static_assert(KV == 0 || 4 * KS == 16 * RT, "VALU tail needs 4*KS == 16*RT");
    __shared__ double sA[2][ROWS][LDA];
    __shared__ double sY[2][16];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lc = lane & 15;
    const int lk = lane >> 4;
    const int K = a.K, N = a.N;

    int64_t chain[CT];
    bool cvalid[CT];
    double th[CT][KS];
    double thv[CT][KV > 0 ? KV : 1], gv[CT][KV > 0 ? KV : 1];
    double tau[CT];
    v4d G[CT][RT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        chain[c] = ((int64_t)blockIdx.x * 4 + wave) * (16 * CT) + 16 * c + lc;
        cvalid[c] = chain[c] < a.C;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 4 * s + lk;
            th[c][s] = (cvalid[c] && k < K) ? a.theta[chain[c] * K + k] : 0.0;
        }
#pragma unroll
        for (int v = 0; v < KV; ++v) {
            thv[c][v] = (cvalid[c] && KB + v < K) ? a.theta[chain[c] * K + KB + v] : 0.0;
            gv[c][v] = 0.0;
        }
        tau[c] = cvalid[c] ? (a.tau_chain ? a.tau_chain[chain[c]] : a.tau) : 0.0;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) G[c][rt] = (v4d){0.0, 0.0, 0.0, 0.0};
    }

    const int t0 = blockIdx.y * a.tiles_per_split;
    int t1 = t0 + a.tiles_per_split;
    const int ntiles = N / 16;
    if (t1 > ntiles) t1 = ntiles;

    // staging: thread (srow, scol) moves A[16*pass + srow][16 t + scol]; its offset inside
    // the tile's column block is fixed, the tile pointer is uniform
    const int srow = tid >> 4, scol = tid & 15;
    unsigned voff[PASSES];                     // BYTE offsets (K N <= 2^28 entries: 31 bits)
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int k = 16 * ps + srow;
        voff[ps] = ((unsigned)(k < K ? k : K - 1) * (unsigned)N + (unsigned)scol) * 8u;
    }
    const unsigned yoff = (unsigned)scol * 8u;
    // buffer resources over A and ys: an address is resource (SGPRs) + per-thread byte
    // offset (one VGPR, fixed) + tile offset (an SGPR the scalar unit advances) -- the
    // prefetch costs no VALU instruction (global_load needs a 64-bit VGPR address per
    // load, which loop strength reduction keeps as four induction pointers).
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
        (void *)a.A, 0, (unsigned)((int64_t)K * N * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(
        (void *)a.ys, 0, (unsigned)((int64_t)N * 8), 0x00020000);
    double pre[PASSES];
    double prey = 0.0;
    auto fetch = [&](int t) {
        const int soff = t * 128;               // 16 doubles per tile
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps)
            pre[ps] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rA, voff[ps], soff, 0));
        prey = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rY, yoff, soff, 0));
    };
    auto stash = [&](auto bc) {
        constexpr int buf = decltype(bc)::value;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int k = 16 * ps + srow;
            if (16 * ps + 15 < ROWS || k < ROWS) sA[buf][k][scol] = pre[ps];
        }
        if (tid < 16) sY[buf][tid] = prey;
    };
    auto tile = [&](auto bc, int t) {
        constexpr int buf = decltype(bc)::value;
        fetch(t + 1 < t1 ? t + 1 : t);         // in flight during the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        // forward: M^T[n][c] = sum_k A[k][n] theta[c][k]
        v4d acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const double av = sA[buf][4 * s + lk][lc];
#pragma unroll
            for (int c = 0; c < CT; ++c)
                acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, th[c][s], acc[c], 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < KV; ++v)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double av = sA[buf][KB + v][lk + 4 * r];
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    acc[c][r] = __builtin_fma(thv[c][v], av, acc[c][r]);
            }
        // error-model gradient in place: r[n][c] = (mock - y[n]) * tau_c
        double rr[CT][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double yv = sY[buf][lk + 4 * r];
#pragma unroll
            for (int c = 0; c < CT; ++c) rr[c][r] = (acc[c][r] - yv) * tau[c];
        }
        // backward: G^T[i][c] += sum_n A[i][n] r[n][c]
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double av = sA[buf][16 * rt + lc][4 * s + lk];
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    G[c][rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, rr[c][s], G[c][rt], 0, 0, 0);
            }
#pragma unroll
        for (int v = 0; v < KV; ++v)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double av = sA[buf][KB + v][lk + 4 * r];
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    gv[c][v] = __builtin_fma(av, rr[c][r], gv[c][v]);
            }
        if (t + 1 < t1) stash(BufC<buf ^ 1>());
        __syncthreads();
    };

    if (t0 < t1) {
        fetch(t0);
        stash(BufC<0>());
    }
    __syncthreads();
    int t = t0;
    for (; t + 1 < t1; t += 2) {
        tile(BufC<0>(), t);
        tile(BufC<1>(), t + 1);
    }
    if (t < t1) tile(BufC<0>(), t);

#pragma unroll
    for (int c = 0; c < CT; ++c) {
#pragma unroll
        for (int v = 0; v < KV; ++v) {
            gv[c][v] = gv[c][v] + shfl_xor_f64(gv[c][v], 16);
            gv[c][v] = gv[c][v] + shfl_xor_f64(gv[c][v], 32);
        }
        if (cvalid[c]) {
            double *dst = a.part + ((int64_t)blockIdx.y * a.C + chain[c]) * K;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * rt + lk + 4 * r;
                    if (i < K) dst[i] = G[c][rt][r];
                }
            if (lk == 0) {
#pragma unroll
                for (int v = 0; v < KV; ++v)
                    if (KB + v < K) dst[KB + v] = gv[c][v];
            }
        }
    }
}

// out[j] = part[0][j] + part[1][j] + ... in split order (deterministic)
__global__ void split_reduce_kernel(const double *part, double *out, int64_t n,
                                    int32_t ns)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    double s = part[j];
    for (int k = 1; k < ns; ++k) s = s + part[(int64_t)k * n + j];
    out[j] = s;
}

// The same sum, followed by the kick it feeds and the drift that follows the kick:
// g = part[0][j] + part[1][j] + ...;  p[j] -= step * g;  q[j] += p[j] * dt
// (hmc.py:116 + 119, 120 + 119 / 122, or 123 alone) -- one launch instead of
// split_reduce_kernel + kick (+ drift) of the per-step tier, the same roundings.
// leap: 1 half kick + drift, 2 kick + drift, 3 half kick.
template <bool FMA>
__global__ void __launch_bounds__(256)
split_reduce_kick_drift_kernel(const double *part, double *q, double *p, const double *dt_chain,
                               double timestep, int64_t n, int32_t K, int32_t ns, int32_t leap)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    double g = part[j];
    for (int k = 1; k < ns; ++k) g = g + part[(int64_t)k * n + j];
    const double d = dt_chain ? dt_chain[j / K] : timestep;
    const double step = (leap == 2) ? d : 0.5 * d;          // "0.5 * timestep" first
    const double pv = p[j];
    const double pn = FMA ? __builtin_fma(-step, g, pv) : pv - step * g;
    p[j] = pn;
    if (leap != 3) {
        const double qv = q[j];
        q[j] = FMA ? __builtin_fma(pn, d, qv) : qv + pn * d;
    }
}

static int grad_ct(int64_t C)
{
    // two 16-chain tiles per wave once there are enough chains to fill the chip (settled
    // sweep of the whole-tile kernel, gpurun_out/r04_grad_sweep2: 4096 chains 63.7 vs 62.9
    // TFLOP/s, 2048 chains 55.0 vs 57.8)
    static int forced = -1;
    if (forced < 0) {
        const char *e = getenv("BINF_POLY_GRAD_CT");      // development aid
        forced = e ? atoi(e) : 0;
    }
    if (forced == 1 || forced == 2) return forced;
    return C >= 4096 ? 2 : 1;
}

static int grad_splits(int64_t C, int64_t N)
{
    // aim at 2 workgroups per CU (2 waves per SIMD keep the matrix pipe fed; fewer, longer
    // workgroups mean fewer partial sums to write and add up); never more splits than data tiles
    const int64_t wgx = (C + 64 * grad_ct(C) - 1) / (64 * grad_ct(C));
    const int64_t ntiles = (N + 15) / 16;
    static int64_t target = 0;
    if (target == 0) {
        const char *e = getenv("BINF_POLY_GRAD_WGS");     // development aid
        target = e ? atoll(e) : 512;             // measured best on MI355X (round 4, settled: 512 / 640 /
                                                 // 768 / 1024 at 1024 .. 8192 chains; 384 starves the chip)
    }
    int64_t ns = (target + wgx - 1) / wgx;
    if (ns > 64) ns = 64;
    if (ns > ntiles) ns = ntiles;
    if (ns < 1) ns = 1;
    return (int)ns;
}

static bool grad_whole_tiles(const GradArgs &a)
{
    // every tile whole and every staging offset inside 32 bits: the trimmed kernel
    static int general = -1;
    if (general < 0) {
        const char *e = getenv("BINF_POLY_GRAD_GENERAL");   // development aid: A/B the two kernels
        general = (e && atoi(e)) ? 1 : 0;
    }
    return !general && a.N >= 16 && a.N % 16 == 0 && (int64_t)a.K * a.N < (1LL << 28);
}

template <int KS, int RT, int KV>
static hipError_t grad_launch(const GradArgs &a, int ct, dim3 grid, hipStream_t st)
{
    if (grad_whole_tiles(a)) {
        if (ct == 2) poly_grad_mfma_full_kernel<KS, RT, KV, 2><<<grid, 256, 0, st>>>(a);
        else         poly_grad_mfma_full_kernel<KS, RT, KV, 1><<<grid, 256, 0, st>>>(a);
    } else {
        if (ct == 2) poly_grad_mfma_kernel<KS, RT, KV, 2><<<grid, 256, 0, st>>>(a);
        else         poly_grad_mfma_kernel<KS, RT, KV, 1><<<grid, 256, 0, st>>>(a);
    }
    return hipGetLastError();
}

}  // namespace binf

using namespace binf;

#define BINF_KMAX_DISPATCH(K, CALL)                                          \
    do {                                                                      \
        if ((K) <= 4) { CALL(4); }                                            \
        else if ((K) <= 8) { CALL(8); }                                       \
        else if ((K) <= 16) { CALL(16); }                                     \
        else if ((K) <= 24) { CALL(24); }                                     \
        else if ((K) <= 33) { CALL(33); }  /* degree 32: BASELINE C3 / C4 */  \
        else if ((K) <= 36) { CALL(36); }                                     \
        else if ((K) <= 48) { CALL(48); }                                     \
        else { CALL(64); }                                                    \
    } while (0)

static int32_t check_poly(const char *what, int64_t C, int64_t K, int64_t N)
{
    if (C < 0 || K < 1 || N < 0)
        return fail(BINF_E_ARG, "%s: need C>=0, K>=1, N>=0", what);
    if (K > 64)
        return fail(BINF_E_UNSUPPORTED, "%s: K=%lld coefficients > 64 not covered by the native polynomial kernels", what, (long long)K);
    if (C > 65535 * 64LL || N > 0x7fffffffLL)
        return fail(BINF_E_UNSUPPORTED, "%s: too large", what);
    return 0;
}

extern "C" int32_t binf_poly_forward_f64(const double *coeffs, const double *xs,
                                         double *out, int64_t C, int64_t K,
                                         int64_t N, void *stream)
{
    int32_t rc = check_poly("poly_forward", C, K, N);
    if (rc) return rc;
    if (C == 0 || N == 0) return 0;
    if (!coeffs || !xs || !out) return fail(BINF_E_ARG, "poly_forward: null buffer");
    int64_t bx = (N + 255) / 256;
    if (bx > 256) bx = 256;
    hipStream_t st = (hipStream_t)stream;
    for (int64_t c0 = 0; c0 < C; c0 += 65535) {            // gridDim.y limit
        const int64_t cn = (C - c0 < 65535) ? C - c0 : 65535;
        PolyArgs a;
        a.theta = coeffs + c0 * K; a.xs = xs; a.ys = nullptr; a.K = (int32_t)K;
        dim3 grid((unsigned)bx, (unsigned)cn);
        double *o = out + c0 * N;
#define CALL(KM) poly_forward_kernel<KM><<<grid, 256, 0, st>>>(a, o, cn, N)
        BINF_KMAX_DISPATCH(K, CALL);
#undef CALL
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "poly_forward launch");
    return 0;
}

extern "C" int32_t binf_gauss_err_grad_f64(const double *mock, const double *ys,
                                           double precision,
                                           const double *precision_chain,
                                           double *out, int64_t C, int64_t N,
                                           void *stream)
{
    if (C < 0 || N < 0) return fail(BINF_E_ARG, "gauss_err_grad: negative size");
    if (C == 0 || N == 0) return 0;
    if (!mock || !ys || !out) return fail(BINF_E_ARG, "gauss_err_grad: null buffer");
    int64_t bx = (N + 255) / 256;
    if (bx > 256) bx = 256;
    for (int64_t c0 = 0; c0 < C; c0 += 65535) {            // gridDim.y limit
        const int64_t cn = (C - c0 < 65535) ? C - c0 : 65535;
        gauss_err_grad_kernel<<<dim3((unsigned)bx, (unsigned)cn), 256, 0, (hipStream_t)stream>>>(
            mock + c0 * N, ys, precision, precision_chain ? precision_chain + c0 : nullptr,
            out + c0 * N, N);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gauss_err_grad launch");
    return 0;
}

extern "C" int32_t binf_gauss_err_logp_f64(const double *mock, const double *ys,
                                           double precision,
                                           const double *precision_chain,
                                           double *out, int64_t C, int64_t N,
                                           void *stream)
{
    // chi^2 in np.sum order, then -0.5*chi2*tau + N*0.5*log(tau)
    int32_t rc = binf_row_sumsq_diff_f64(mock, ys, nullptr, out, C, N, 1.0, stream);
    if (rc || C == 0) return rc;
    gauss_logp_finish_kernel<<<dim3((unsigned)((C + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
        out, precision, precision_chain, out, C, (double)N);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gauss_err_logp launch");
    return 0;
}

extern "C" int32_t binf_poly_gauss_logp_f64(const double *coeffs, const double *xs,
                                            const double *ys, double precision,
                                            const double *precision_chain,
                                            double *out, int64_t C, int64_t K,
                                            int64_t N, void *stream)
{
    int32_t rc = check_poly("poly_gauss_logp", C, K, N);
    if (rc) return rc;
    if (C == 0) return 0;
    if (!coeffs || ((!xs || !ys) && N > 0) || !out)
        return fail(BINF_E_ARG, "poly_gauss_logp: null buffer");
    PolyArgs a;
    a.theta = coeffs; a.xs = xs; a.ys = ys; a.K = (int32_t)K;
    hipStream_t st = (hipStream_t)stream;
    // up to 1024 data points a row is one lane group of a wave (8 .. 64 lanes; the
    // example's 20 points: 8 rows per wave instead of a 256-thread workgroup each,
    // 9.7 -> 3 us at 8192 chains); beyond, one workgroup per chain.  Same np.sum order.
    GaussFinish fin;                // the error model's log-prob is the reduction's epilogue
    fin.on = 1; fin.minus = nullptr; fin.tau = precision; fin.tau_chain = precision_chain; fin.n_data = (double)N;
#define CALL(KM) rc = row_reduce_launch<ResidSqMake<KM>, PolyArgs>(a, C, N, 1.0, out, st, false, "poly_gauss_logp", 0, false, &fin)
    BINF_KMAX_DISPATCH(K, CALL);
#undef CALL
    return rc;
}

extern "C" int32_t binf_poly_gauss_logp_memo_f64(const double *coeffs, const double *xs,
                                                 const double *ys, double precision,
                                                 const double *precision_chain, double *out,
                                                 double *memo_coeffs, double *memo_chi2,
                                                 uint8_t *skip, int64_t C, int64_t K, int64_t N,
                                                 void *stream)
{
    int32_t rc = check_poly("poly_gauss_logp_memo", C, K, N);
    if (rc) return rc;
    if (C == 0) return 0;
    if (!coeffs || ((!xs || !ys) && N > 0) || !out || !memo_coeffs || !memo_chi2 || !skip)
        return fail(BINF_E_ARG, "poly_gauss_logp_memo: null buffer");
    hipStream_t st = (hipStream_t)stream;
    // every refusal of the reduction BEFORE the memo is touched (its check kernel rewrites
    // the stored coefficients of a missed row; their chi^2 is stored by the reduction)
    rc = row_reduce_check(C, N, "poly_gauss_logp_memo");
    if (rc) return rc;
    // which chains still have the coefficients their stored chi^2 belongs to (rowsum.hpp)
    rc = row_memo_check(coeffs, memo_coeffs, skip, C, K, st, "poly_gauss_logp_memo check launch");
    if (rc) return rc;
    PolyArgs a;
    a.theta = coeffs; a.xs = xs; a.ys = ys; a.K = (int32_t)K;
    GaussFinish fin;
    fin.on = 1; fin.minus = nullptr; fin.tau = precision; fin.tau_chain = precision_chain; fin.n_data = (double)N;
    // one workgroup per chain whatever N is (force_block): it is the kernel that honours the memo
#define CALL(KM) rc = row_reduce_launch<ResidSqMake<KM>, PolyArgs>(a, C, N, 1.0, out, st, true, "poly_gauss_logp_memo", 0, false, &fin, skip, memo_chi2)
    BINF_KMAX_DISPATCH(K, CALL);
#undef CALL
    return rc;
}

extern "C" int32_t binf_gamma_logp_f64(const double *precision, double shape, double rate,
                                       double *out, int64_t C, void *stream)
{
    if (C < 0) return fail(BINF_E_ARG, "gamma_logp: negative size");
    if (C == 0) return 0;
    if (!precision || !out) return fail(BINF_E_ARG, "gamma_logp: null buffer");
    gamma_logp_kernel<<<dim3((unsigned)((C + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
        precision, shape - 1.0, rate, out, C);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gamma_logp launch");
    return 0;
}

static hipError_t grad_dispatch(const GradArgs &a, int64_t K, int ct, dim3 grid, hipStream_t st)
{
    hipError_t e;
    if (K <= 4)       e = grad_launch<1, 1, 0>(a, ct, grid, st);
    else if (K <= 8)  e = grad_launch<2, 1, 0>(a, ct, grid, st);
    else if (K <= 16) e = grad_launch<4, 1, 0>(a, ct, grid, st);
    else if (K == 17) e = grad_launch<4, 1, 1>(a, ct, grid, st);
    else if (K == 18) e = grad_launch<4, 1, 2>(a, ct, grid, st);
    else if (K <= 32) e = grad_launch<8, 2, 0>(a, ct, grid, st);
    else if (K == 33) e = grad_launch<8, 2, 1>(a, ct, grid, st);
    else if (K == 34) e = grad_launch<8, 2, 2>(a, ct, grid, st);
    else if (K <= 36) e = grad_launch<9, 3, 0>(a, ct, grid, st);
    else if (K <= 48) e = grad_launch<12, 3, 0>(a, ct, grid, st);
    else if (K == 49) e = grad_launch<12, 3, 1>(a, ct, grid, st);
    else if (K == 50) e = grad_launch<12, 3, 2>(a, ct, grid, st);
    else              e = grad_launch<16, 4, 0>(a, ct, grid, st);
    return e;
}

extern "C" int64_t binf_poly_gauss_grad_workspace_bytes(int64_t C, int64_t K, int64_t N)
{
    if (C <= 0 || K <= 0 || N <= 0) return 0;
    const int ns = grad_splits(C, N);
    return ns > 1 ? (int64_t)ns * C * K * (int64_t)sizeof(double) : 0;
}

extern "C" int32_t binf_poly_gauss_grad_f64(const double *coeffs, const double *design,
                                            const double *ys, double precision,
                                            const double *precision_chain,
                                            double *out, void *workspace,
                                            int64_t workspace_bytes, int64_t C,
                                            int64_t K, int64_t N, void *stream)
{
    int32_t rc = check_poly("poly_gauss_grad", C, K, N);
    if (rc) return rc;
    if (C == 0) return 0;
    if (!coeffs || !out || ((!design || !ys) && N > 0))
        return fail(BINF_E_ARG, "poly_gauss_grad: null buffer");
    // The data range is summed in ns pieces (partial sums reduced in a fixed
    // order); ns follows from (C, N) alone -- binf_poly_gauss_grad_workspace_bytes
    // reports the scratch it needs -- so a call is deterministic, but the SAME
    // chain evaluated in batches of different size may differ at rounding level
    // (as the reference's BLAS contraction does between builds); a missing or
    // short workspace is an error, never a silent change of summation order.
    const int ns = grad_splits(C, N);
    const int64_t need = ns > 1 ? (int64_t)ns * C * K * (int64_t)sizeof(double) : 0;
    if (need > 0 && (!workspace || workspace_bytes < need))
        return fail(BINF_E_ARG, "poly_gauss_grad: needs %lld bytes of workspace "
                    "(binf_poly_gauss_grad_workspace_bytes), got %lld",
                    (long long)need, (long long)workspace_bytes);
    const int ntiles = (int)((N + 15) / 16);
    GradArgs a;
    a.theta = coeffs; a.A = design; a.ys = ys; a.tau_chain = precision_chain;
    a.tau = precision; a.C = C; a.K = (int32_t)K; a.N = (int32_t)N;
    a.tiles_per_split = (ntiles + ns - 1) / ns;
    if (a.tiles_per_split < 1) a.tiles_per_split = 1;
    a.part = ns > 1 ? (double *)workspace : out;
    const int ct = grad_ct(C);
    dim3 grid((unsigned)((C + 64 * ct - 1) / (64 * ct)), (unsigned)ns);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    e = grad_dispatch(a, K, ct, grid, st);
    if (e != hipSuccess) return hip_fail(e, "poly_gauss_grad launch");
    if (ns > 1) {
        const int64_t n = C * K;
        split_reduce_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(
            (const double *)workspace, out, n, ns);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "poly_gauss_grad reduce launch");
    }
    return 0;
}

extern "C" int64_t binf_poly_leapfrog_workspace_bytes(int64_t C, int64_t K, int64_t N)
{
    if (C <= 0 || K <= 0 || N <= 0) return 0;
    return (int64_t)grad_splits(C, N) * C * K * (int64_t)sizeof(double);
}

extern "C" int32_t binf_poly_leapfrog_f64(double *q, double *p, const double *design,
                                          const double *ys, double precision,
                                          const double *precision_chain, void *workspace,
                                          int64_t workspace_bytes, int64_t C, int64_t K,
                                          int64_t N, double timestep, const double *dt_chain,
                                          int32_t nsteps, int32_t mode, void *stream)
{
    int32_t rc = check_poly("poly_leapfrog", C, K, N);
    if (rc) return rc;
    if (nsteps < 1) return fail(BINF_E_ARG, "poly_leapfrog: nsteps >= 1 required");
    if (mode != BINF_MODE_EXACT && mode != BINF_MODE_FMA)
        return fail(BINF_E_ARG, "poly_leapfrog: unknown mode %d", mode);
    if (C == 0) return 0;
    if (N < 1) return fail(BINF_E_UNSUPPORTED, "poly_leapfrog: no data points");
    if (!q || !p || !design || !ys) return fail(BINF_E_ARG, "poly_leapfrog: null buffer");
    const int64_t need = binf_poly_leapfrog_workspace_bytes(C, K, N);
    if (!workspace || workspace_bytes < need)
        return fail(BINF_E_ARG, "poly_leapfrog: needs %lld bytes of workspace "
                    "(binf_poly_leapfrog_workspace_bytes), got %lld",
                    (long long)need, (long long)workspace_bytes);
    const int ns = grad_splits(C, N);
    const int ntiles = (int)((N + 15) / 16);
    const int ct = grad_ct(C);
    GradArgs a;
    a.theta = q; a.A = design; a.ys = ys; a.tau_chain = precision_chain; a.tau = precision;
    a.C = C; a.K = (int32_t)K; a.N = (int32_t)N;
    a.tiles_per_split = (ntiles + ns - 1) / ns;
    if (a.tiles_per_split < 1) a.tiles_per_split = 1;
    a.part = (double *)workspace;
    dim3 grid((unsigned)((C + 64 * ct - 1) / (64 * ct)), (unsigned)ns);
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = C * K;
    const dim3 rgrid((unsigned)((n + 255) / 256));
    // hmc.py:116-123: half kick, (nsteps - 1) x [drift, kick], drift, half kick -- per
    // gradient one MFMA launch and one launch for partial sums + kick + drift
    for (int l = 0; l <= nsteps; ++l) {
        const int leap = (l == 0) ? 1 : (l == nsteps ? 3 : 2);
        hipError_t e = grad_dispatch(a, K, ct, grid, st);
        if (e != hipSuccess) return hip_fail(e, "poly_leapfrog gradient launch");
        if (mode == BINF_MODE_FMA)
            split_reduce_kick_drift_kernel<true><<<rgrid, 256, 0, st>>>(
                (const double *)workspace, q, p, dt_chain, timestep, n, (int32_t)K, ns, leap);
        else
            split_reduce_kick_drift_kernel<false><<<rgrid, 256, 0, st>>>(
                (const double *)workspace, q, p, dt_chain, timestep, n, (int32_t)K, ns, leap);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "poly_leapfrog kick launch");
    }
    return 0;
}

extern "C" int32_t binf_gamma_precision_update_f64(const double *g, const double *lp_unit,
                                                   double prior_rate, double *out,
                                                   int64_t C, void *stream)
{
    if (C < 0) return fail(BINF_E_ARG, "gamma_precision_update: negative size");
    if (C == 0) return 0;
    if (!g || !lp_unit || !out) return fail(BINF_E_ARG, "gamma_precision_update: null buffer");
    gamma_update_kernel<<<dim3((unsigned)((C + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
        g, lp_unit, prior_rate, out, C);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gamma_precision_update launch");
    return 0;
}

// The generic chain-rule contraction of the Likelihood plug-in surface and the
// Posterior's term sums -- the two steps of the generic tier that ran as library
// BLAS / elementwise calls before:
//
//   Likelihood._evaluate_gradient   binf/pdf/likelihoods.py:148-155
//       return dfm.dot(emgrad)      J [n_params x n_data] . dE/dmock [n_data]
//   Posterior._evaluate_log_prob    binf/pdf/posteriors.py:147-151
//       numpy.sum([c.log_prob(...) for c in components])   (a short list: sequential)
//   Posterior._evaluate_gradient    binf/pdf/posteriors.py:173-187
//       sum(f.gradient(...) for f in components)
//
// binf_jacobian_contract_f64 serves any forward model without a fused kernel:
//   shared Jacobian  J [K x N] (the same for every chain: linear models)  -> f64 MFMA,
//   per-chain Jacobian J [C x K x N]                                       -> streamed GEMV.
// The reference's BLAS order is not reproducible, so the contraction is held to
// 1e-10 of sum_n |J||r| (tests/poly_bounds.py) -- but its own summation order is
// FIXED by (K, N) alone: a call is deterministic and independent of the batch.
// gfx950, wave64.
#include "common.hpp"

namespace binf {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int JT_N = 64;          // data points per staged tile
constexpr int JT_LD = JT_N + 1;   // padded LDS row (doubles)

// ---- shared Jacobian: out[c][k] = sum_n r[c][n] J[k][n] ------------------------
// A workgroup owns 16 chains x 16 RT rows of J.  Per tile of 64 data points the
// rows of J and of r are staged in LDS with coalesced 512-byte row segments (the next
// tile prefetched into registers, two LDS images, one barrier per tile); wave
// w multiplies the tile's data points 16 w .. 16 w + 15 (four k-steps of the
// 16x16x4 f64 MFMA: A[m = row of J][k = n], B[k = n][col = chain]).  The four
// waves' accumulators are joined at the end in the order (w0 + w1) + (w2 + w3).
template <int RT>
__global__ void __launch_bounds__(256)
jac_shared_mfma_kernel(const double *__restrict__ r, const double *__restrict__ J,
                       double *__restrict__ out, int64_t C, int32_t K, int32_t N)
{
    // two LDS images of a tile (rows of J, rows of r): tile t + 1 is written while tile t is
    // multiplied, ONE barrier per tile; the accumulator exchange at the end reuses image 0
    constexpr int SJ_N = 16 * RT * JT_LD, SACC_N = 4 * 16 * RT * 17;
    __shared__ double sbuf[2][SJ_N > SACC_N ? SJ_N : SACC_N];
    __shared__ double sRb[2][16][JT_LD];
    double (*sAcc)[16 * RT][17] = reinterpret_cast<double (*)[16 * RT][17]>(sbuf[0]);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lc = lane & 15, lk = lane >> 4;
    const int64_t c0 = (int64_t)blockIdx.x * 16;
    const int k0 = blockIdx.y * 16 * RT;
    v4d acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int col = tid & 63, row0 = tid >> 6;              // staging: 4 rows per pass
    // the NEXT tile's rows travel in registers while this tile's products run (a workgroup walks
    // its N / 64 tiles in sequence: without the prefetch every tile paid a full HBM latency)
    double pj[4 * RT], pr[4];
    auto fetch = [&](int n0) {
        const int n = n0 + col;
        const bool nv = n < N;
#pragma unroll
        for (int p = 0; p < 4 * RT; ++p) {
            const int k = k0 + 4 * p + row0;
            pj[p] = (nv && k < K) ? J[(int64_t)k * N + n] : 0.0;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int64_t c = c0 + 4 * p + row0;
            pr[p] = (nv && c < C) ? r[c * N + n] : 0.0;
        }
    };
    auto stage = [&](int b) {
        double (*sJ)[JT_LD] = reinterpret_cast<double (*)[JT_LD]>(sbuf[b]);
#pragma unroll
        for (int p = 0; p < 4 * RT; ++p) sJ[4 * p + row0][col] = pj[p];
#pragma unroll
        for (int p = 0; p < 4; ++p) sRb[b][4 * p + row0][col] = pr[p];
    };
    fetch(0);
    stage(0);
    __syncthreads();
    int b = 0;
    for (int n0 = 0; n0 < N; n0 += JT_N, b ^= 1) {
        fetch(n0 + JT_N);                                   // beyond N: nothing is loaded
        double (*sJ)[JT_LD] = reinterpret_cast<double (*)[JT_LD]>(sbuf[b]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int nn = 16 * wave + 4 * s + lk;
            const double bv = sRb[b][lc][nn];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(sJ[16 * rt + lc][nn], bv, acc[rt], 0, 0, 0);
        }
        stage(b ^ 1);                                       // every wave left that image a barrier ago
        __syncthreads();
    }
    // D layout: row (of J) = lk + 4 i, column (chain) = lc
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) sAcc[wave][16 * rt + lk + 4 * i][lc] = acc[rt][i];
    __syncthreads();
    for (int e = tid; e < 16 * RT * 16; e += 256) {
        const int kr = e >> 4, cc = e & 15;
        const int k = k0 + kr;
        const int64_t c = c0 + cc;
        if (k < K && c < C)
            out[c * K + k] = (sAcc[0][kr][cc] + sAcc[1][kr][cc]) + (sAcc[2][kr][cc] + sAcc[3][kr][cc]);
    }
}

// The same with 32 chains per workgroup of 8 waves: waves 0-3 multiply the tile of J with
// chains 0-15, waves 4-7 with chains 16-31 -- every workgroup streams all of J through L2, so
// twice the chains per workgroup is half of that traffic (at C3's size it is twice the bytes
// of the residuals themselves).  Same order of a chain's sums, same bits.
template <int RT>
__global__ void __launch_bounds__(512)
jac_shared_mfma32_kernel(const double *__restrict__ r, const double *__restrict__ J,
                         double *__restrict__ out, int64_t C, int32_t K, int32_t N)
{
    constexpr int SJ_N = 16 * RT * JT_LD, SACC_N = 8 * 16 * RT * 17;
    __shared__ double sbuf[2 * SJ_N > SACC_N ? 2 * SJ_N : SACC_N];
    __shared__ double sRb[2][32][JT_LD];
    double (*sAcc)[16 * RT][17] = reinterpret_cast<double (*)[16 * RT][17]>(sbuf);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = wave >> 2, w4 = wave & 3;
    const int lc = lane & 15, lk = lane >> 4;
    const int64_t c0 = (int64_t)blockIdx.x * 32;
    const int k0 = blockIdx.y * 16 * RT;
    v4d acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int col = tid & 63, row0 = tid >> 6;              // staging: 8 rows per pass
    double pj[2 * RT], pr[4];
    auto fetch = [&](int n0) {
        const int n = n0 + col;
        const bool nv = n < N;                              // beyond N: nothing is loaded
#pragma unroll
        for (int p = 0; p < 2 * RT; ++p) {
            const int k = k0 + 8 * p + row0;
            pj[p] = (nv && k < K) ? J[(int64_t)k * N + n] : 0.0;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int64_t c = c0 + 8 * p + row0;
            pr[p] = (nv && c < C) ? r[c * N + n] : 0.0;
        }
    };
    auto stage = [&](int b) {
        double (*sJ)[JT_LD] = reinterpret_cast<double (*)[JT_LD]>(sbuf + b * SJ_N);
#pragma unroll
        for (int p = 0; p < 2 * RT; ++p) sJ[8 * p + row0][col] = pj[p];
#pragma unroll
        for (int p = 0; p < 4; ++p) sRb[b][8 * p + row0][col] = pr[p];
    };
    fetch(0);
    stage(0);
    __syncthreads();
    int b = 0;
    for (int n0 = 0; n0 < N; n0 += JT_N, b ^= 1) {
        fetch(n0 + JT_N);
        double (*sJ)[JT_LD] = reinterpret_cast<double (*)[JT_LD]>(sbuf + b * SJ_N);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int nn = 16 * w4 + 4 * s + lk;
            const double bv = sRb[b][16 * half + lc][nn];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(sJ[16 * rt + lc][nn], bv, acc[rt], 0, 0, 0);
        }
        stage(b ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) sAcc[wave][16 * rt + lk + 4 * i][lc] = acc[rt][i];
    __syncthreads();
    for (int e = tid; e < 2 * 16 * RT * 16; e += 512) {
        const int h = e / (16 * RT * 16), kr = (e >> 4) % (16 * RT), cc = e & 15;
        const int k = k0 + kr;
        const int64_t c = c0 + 16 * h + cc;
        if (k < K && c < C)
            out[c * K + k] = (sAcc[4 * h][kr][cc] + sAcc[4 * h + 1][kr][cc]) +
                             (sAcc[4 * h + 2][kr][cc] + sAcc[4 * h + 3][kr][cc]);
    }
}

// ---- per-chain Jacobian: out[c][k] = sum_n J[c][k][n] r[c][n] -------------------
// One workgroup per chain; wave w owns rows k = w, w + 4, ...; lane l adds the
// products of n = l, l + 64, ... in order (coalesced 512-byte row segments), the
// lanes are joined by an xor butterfly: the order depends on N only.
__global__ void __launch_bounds__(256)
jac_batched_kernel(const double *__restrict__ r, const double *__restrict__ J,
                   double *__restrict__ out, int64_t C, int32_t K, int32_t N)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const double *rc = r + c * N;
        for (int k = wave; k < K; k += 4) {
            const double *row = J + (c * K + k) * (int64_t)N;
            double s0 = 0.0, s1 = 0.0;
            int n = lane;
            for (; n + 64 < N; n += 128) {                   // two independent chains per lane
                s0 = __builtin_fma(row[n], rc[n], s0);
                s1 = __builtin_fma(row[n + 64], rc[n + 64], s1);
            }
            if (n < N) s0 = __builtin_fma(row[n], rc[n], s0);
            double s = s0 + s1;
            s = sum8_f64(s);
            s = s + xor8_f64(s);
            s = s + xor16_f64(s, lane);
            s = s + xor32_f64(s, lane);
            if (lane == 0) out[c * K + k] = s;
        }
    }
}

// ---- out[i] = ((t0[i] + t1[i]) + t2[i]) + ...  (sequential, as numpy.sum of a short
// list adds; a term is a device vector or a host scalar) -------------------------
constexpr int SUM_TERMS_MAX = 16;
struct SumTermsArgs {
    const double *ptr[SUM_TERMS_MAX];
    double scalar[SUM_TERMS_MAX];
    uint8_t bcast[SUM_TERMS_MAX];      // ptr[t] is ONE device double, the same for every i
    int32_t T;
};

__global__ void __launch_bounds__(256) sum_terms_kernel(const SumTermsArgs a, double *out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        double s = a.ptr[0] ? a.ptr[0][a.bcast[0] ? 0 : i] : a.scalar[0];
        for (int t = 1; t < a.T; ++t)
            s = s + (a.ptr[t] ? a.ptr[t][a.bcast[t] ? 0 : i] : a.scalar[t]);
        out[i] = s;
    }
}

// development aid: BINF_JAC_CHAINS=16 keeps the 16-chain workgroups (A/B, bit-equality tests)
static bool jac_force16()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("BINF_JAC_CHAINS");
        v = (e && atoi(e) == 16) ? 1 : 0;
    }
    return v == 1;
}

}  // namespace binf

using namespace binf;

extern "C" int32_t binf_jacobian_contract_f64(const double *jacobian, const double *emgrad,
                                              double *out, int64_t C, int64_t K, int64_t N,
                                              int32_t batched, void *stream)
{
    if (C < 0 || K < 0 || N < 0) return fail(BINF_E_ARG, "jacobian_contract: negative size");
    if (C == 0 || K == 0) return 0;
    if (!out || (N > 0 && (!jacobian || !emgrad)))
        return fail(BINF_E_ARG, "jacobian_contract: null buffer");
    if (K > 0x7fffffff || N > 0x7fffffff)
        return fail(BINF_E_UNSUPPORTED, "jacobian_contract: K, N must fit 32 bits");
    hipStream_t st = (hipStream_t)stream;
    if (batched) {
        int64_t blocks = C < 65536 ? C : 65536;
        jac_batched_kernel<<<dim3((unsigned)blocks), 256, 0, st>>>(emgrad, jacobian, out, C, (int32_t)K,
                                                                    (int32_t)N);
    } else {
        const int64_t ctiles = (C + 15) / 16;
        if (ctiles > 0x7fffffffLL) return fail(BINF_E_UNSUPPORTED, "jacobian_contract: too many chains");
        // enough chains for a 32-chain workgroup on every CU, and enough data points for J's
        // traffic to matter: the 8-wave kernel (half the reads of J)
        const int64_t ctiles32 = (C + 31) / 32;
        if (ctiles32 >= 256 && N >= 1024 && K <= 64 && !jac_force16()) {
            const dim3 g32((unsigned)ctiles32, 1);
            if (K <= 16) jac_shared_mfma32_kernel<1><<<g32, 512, 0, st>>>(emgrad, jacobian, out, C, (int32_t)K, (int32_t)N);
            else if (K <= 32) jac_shared_mfma32_kernel<2><<<g32, 512, 0, st>>>(emgrad, jacobian, out, C, (int32_t)K, (int32_t)N);
            else if (K <= 48) jac_shared_mfma32_kernel<3><<<g32, 512, 0, st>>>(emgrad, jacobian, out, C, (int32_t)K, (int32_t)N);
            else jac_shared_mfma32_kernel<4><<<g32, 512, 0, st>>>(emgrad, jacobian, out, C, (int32_t)K, (int32_t)N);
        } else
        // up to 64 rows of J per workgroup (the residual tile is read once for them)
        if (K <= 16)
            jac_shared_mfma_kernel<1><<<dim3((unsigned)ctiles, 1), 256, 0, st>>>(emgrad, jacobian, out, C, (int32_t)K, (int32_t)N);
        else if (K <= 32)
            jac_shared_mfma_kernel<2><<<dim3((unsigned)ctiles, 1), 256, 0, st>>>(emgrad, jacobian, out, C, (int32_t)K, (int32_t)N);
        else if (K <= 48)
            jac_shared_mfma_kernel<3><<<dim3((unsigned)ctiles, 1), 256, 0, st>>>(emgrad, jacobian, out, C, (int32_t)K, (int32_t)N);
        else {
            const int64_t kblocks = (K + 63) / 64;
            if (kblocks > 65535) return fail(BINF_E_UNSUPPORTED, "jacobian_contract: too many parameters");
            jac_shared_mfma_kernel<4><<<dim3((unsigned)ctiles, (unsigned)kblocks), 256, 0, st>>>(emgrad, jacobian, out, C, (int32_t)K, (int32_t)N);
        }
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "jacobian_contract launch");
    return 0;
}

static int32_t sum_terms_launch(const char *what, const double *const *terms, const double *scalars,
                                const uint8_t *broadcast, int32_t n_terms, double *out, int64_t n,
                                void *stream)
{
    if (n_terms < 1 || n_terms > SUM_TERMS_MAX)
        return fail(BINF_E_ARG, "%s: 1 .. %d terms, got %d", what, SUM_TERMS_MAX, n_terms);
    if (n < 0) return fail(BINF_E_ARG, "%s: negative size", what);
    if (n == 0) return 0;
    if (!terms || !out) return fail(BINF_E_ARG, "%s: null pointer", what);
    SumTermsArgs a;
    a.T = n_terms;
    for (int t = 0; t < SUM_TERMS_MAX; ++t) a.bcast[t] = 0;
    for (int t = 0; t < n_terms; ++t) {
        a.ptr[t] = terms[t];
        if (!terms[t] && !scalars) return fail(BINF_E_ARG, "%s: term %d has neither a vector nor a scalar", what, t);
        a.scalar[t] = scalars ? scalars[t] : 0.0;
        a.bcast[t] = (broadcast && broadcast[t]) ? 1 : 0;
        // a broadcast term overlapping out would be overwritten by element 0's sum while
        // other elements still read it
        if (a.bcast[t] && terms[t] && overlap_f64(terms[t], 1, out, n))
            return fail(BINF_E_ALIAS, "%s: broadcast term %d lies inside out", what, t);
    }
    int64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    sum_terms_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(a, out, n);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "sum_terms launch");
    return 0;
}

extern "C" int32_t binf_sum_terms_f64(const double *const *terms, const double *scalars,
                                      int32_t n_terms, double *out, int64_t n, void *stream)
{
    return sum_terms_launch("sum_terms", terms, scalars, nullptr, n_terms, out, n, stream);
}

extern "C" int32_t binf_sum_terms_bcast_f64(const double *const *terms, const double *scalars,
                                            const uint8_t *broadcast, int32_t n_terms, double *out,
                                            int64_t n, void *stream)
{
    return sum_terms_launch("sum_terms_bcast", terms, scalars, broadcast, n_terms, out, n, stream);
}

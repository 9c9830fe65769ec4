// Generic per-step tier: the pieces of HMCSampler.sample() as separate
// chain-batched kernels, for posteriors whose gradient is evaluated by other
// code (any AbstractBinfPDF-shaped plug-in).  gfx950, wave64.
//
// Reference lines replaced: binf/samplers/hmc.py:116-123 (kick / drift),
// :148,150 (energy reductions), :151-164 (accept, adapt, select) and the
// TestHO gradient binf/pdf/__init__.py:191.
#include "common.hpp"

namespace binf {

// ---------------------------------------------------------------------------
// row reductions in numpy's pairwise order
// ---------------------------------------------------------------------------
enum { OP_SUM = 0, OP_SUMSQ = 1, OP_SUMSQ_SHIFT = 2 };

template <int OP>
__device__ inline double red_elem(double x, double shift)
{
    if (OP == OP_SUM) return x;
    if (OP == OP_SUMSQ) return x * x;
    const double d = x - shift;
    return d * d;
}

// Sum of one leaf (<=128 elements at `a`) by the 8 lanes of a group; every
// lane of the wave must call it.  Returns the leaf sum in all 8 lanes.
template <int OP>
__device__ inline double leaf_sum(const double *a, int n, int lane, double shift,
                                  bool active)
{
    const int j = lane & 7;
    const int T = (n >= 8) ? (n >> 3) : 0;
    const int rem = (n >= 8) ? (n & 7) : n;
    double r = 0.0;
    if (active && T > 0) {
        r = red_elem<OP>(a[j], shift);
        for (int t = 1; t < T; ++t) r = r + red_elem<OP>(a[8 * t + j], shift);
    }
    r = r + shfl_xor_f64(r, 1);
    r = r + shfl_xor_f64(r, 2);
    r = r + shfl_xor_f64(r, 4);
    double res = (T > 0) ? r : -0.0;
    double tail = 0.0;
    if (active && j < rem) tail = red_elem<OP>(a[8 * T + j], shift);
    const int leafbase = lane & ~7;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const double v = shfl_f64(tail, leafbase + i);
        const double s = res + v;
        res = (i < rem) ? s : res;
    }
    return res;
}

struct RowSumArgs {
    const double *x;
    double *out;
    int64_t C;
    int32_t D;
    int32_t H;
    double shift;
    double scale;
};

// H <= 3: G = 8<<H lanes of one wave per row, 64/G rows per wave.
template <int OP>
__global__ void __launch_bounds__(256) row_sum_wave_kernel(const RowSumArgs a)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int H = a.H;
    const int lg = 3 + H;
    const int slot = lane & ((1 << lg) - 1);
    const int64_t row_raw = (wave << (6 - lg)) + (lane >> lg);
    const bool valid = row_raw < a.C;
    const int64_t row = valid ? row_raw : a.C - 1;
    const Leaf L = pairwise_leaf(a.D, H, slot >> 3);
    double res = leaf_sum<OP>(a.x + row * (int64_t)a.D + L.off, L.len, lane,
                              a.shift, true);
    for (int l = 0; l < H; ++l) {
        const double o = shfl_xor_f64(res, 8 << l);
        const double s = res + o;
        res = (L.depth >= H - l) ? s : res;
    }
    if (valid && slot == 0) a.out[row] = a.scale * (0.0 + res);
}

// Any D: one 256-thread workgroup per row.  numpy's buffered reduction feeds
// the pairwise loop NPY_BUFSIZE = 8192 elements at a time and adds the chunk
// sums up one after the other; a chunk's tree has height <= 6 (64 leaves), so
// its leaf sums and the leaf tree fit in 64 LDS slots.
constexpr int NPY_BUFSIZE = 8192;

template <int OP>
__global__ void __launch_bounds__(256) row_sum_block_kernel(const RowSumArgs a)
{
    __shared__ double S[64];
    __shared__ int dep[64];
    const int H = a.H;                       // height for min(D, 8192) elements
    const int npaths = 1 << H;
    const int lane = threadIdx.x & 63;
    const int group = threadIdx.x >> 3;      // 32 groups of 8 lanes
    const int64_t row = blockIdx.x;
    const double *x = a.x + row * (int64_t)a.D;
    double total = 0.0;                      // the reduction's identity
    for (int cbase = 0; cbase == 0 || cbase < a.D; cbase += NPY_BUFSIZE) {
        const int n = (a.D - cbase < NPY_BUFSIZE) ? a.D - cbase : NPY_BUFSIZE;
        for (int base = 0; base < npaths; base += 32) {
            const int path = base + group;
            const bool act = path < npaths;
            const Leaf L = pairwise_leaf(n, H, act ? path : 0);
            const double s = leaf_sum<OP>(x + cbase + L.off, L.len, lane,
                                          a.shift, act);
            if (act && (lane & 7) == 0) {
                S[path] = s;
                dep[path] = L.depth;
            }
        }
        __syncthreads();
        for (int l = 0; l < H; ++l) {
            double v = 0.0;
            const int p = threadIdx.x;
            if (p < npaths) {
                const double mine = S[p];
                v = (dep[p] >= H - l) ? mine + S[p ^ (1 << l)] : mine;
            }
            __syncthreads();
            if (p < npaths) S[p] = v;
            __syncthreads();
        }
        total = total + S[0];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.out[row] = a.scale * total;
}

// ---------------------------------------------------------------------------
// elementwise leapfrog pieces
// ---------------------------------------------------------------------------
struct EwArgs {
    double *y;
    const double *x;
    const double *dt_chain;
    double timestep;
    double k;
    double x0;
    int64_t C;
    int64_t D;
    int32_t half;
};

enum { EW_KICK = 0, EW_DRIFT = 1, EW_GAUSS_GRAD = 2 };

template <int KIND, bool FMA>
__global__ void __launch_bounds__(256) ew_kernel(const EwArgs a)
{
    const int64_t c = blockIdx.y;
    double dt = a.dt_chain ? a.dt_chain[c] : a.timestep;
    if (KIND == EW_KICK && a.half) dt = 0.5 * dt;       // "0.5 * timestep" first
    double *y = a.y + c * a.D;
    const double *x = a.x + c * a.D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.D;
         i += (int64_t)gridDim.x * 256) {
        if (KIND == EW_KICK)            // p -= dt * grad      hmc.py:116,120,123
            y[i] = FMA ? __builtin_fma(-dt, x[i], y[i]) : y[i] - dt * x[i];
        else if (KIND == EW_DRIFT)      // q += p * dt         hmc.py:119,122
            y[i] = FMA ? __builtin_fma(x[i], dt, y[i]) : y[i] + x[i] * dt;
        else                            // k * (x - x0)        pdf/__init__.py:191
            y[i] = a.k * (x[i] - a.x0);
    }
}

// ---------------------------------------------------------------------------
// Metropolis accept, step-size adaption, select           hmc.py:151-164,188-191
// ---------------------------------------------------------------------------
struct AcceptArgs {
    const double *q_prop;
    const double *q_old;
    const double *e_before;
    const double *e_after;
    const double *u;
    double *q_out;
    uint8_t *accepted;
    int64_t *n_accepted;
    double *dt_chain;
    double uprate;
    double downrate;
    int64_t C;
    int64_t D;
    int32_t adapt;
};

__global__ void __launch_bounds__(256) accept_select_kernel(const AcceptArgs a)
{
    const int64_t c = blockIdx.x;
    double x = -(a.e_after[c] - a.e_before[c]);
    x = (x < -308.0) ? -308.0 : x;
    x = (x > 709.0) ? 709.0 : x;
    const bool acc = a.u[c] < exp(x);
    const double *src = acc ? a.q_prop : a.q_old;
    double *dst = a.q_out + c * a.D;
    if (dst != src + c * a.D)
        for (int64_t i = threadIdx.x; i < a.D; i += 256) dst[i] = src[c * a.D + i];
    if (threadIdx.x == 0) {
        a.accepted[c] = acc ? 1 : 0;
        if (a.n_accepted && acc) a.n_accepted[c] += 1;
        if (a.adapt) {
            const double dt = a.dt_chain[c];
            a.dt_chain[c] = acc ? dt * a.uprate : dt * a.downrate;
        }
    }
}

template <int OP>
static int32_t row_sum_launch(const RowSumArgs &a, hipStream_t st)
{
    if (a.H <= 3) {
        const int64_t rows_per_wave = 64 >> (3 + a.H);
        const int64_t waves = (a.C + rows_per_wave - 1) / rows_per_wave;
        const int64_t blocks = (waves + 3) / 4;
        if (blocks > 0x7fffffffLL) return fail(BINF_E_UNSUPPORTED, "row_sum: too many rows");
        row_sum_wave_kernel<OP><<<dim3((unsigned)blocks), 256, 0, st>>>(a);
    } else {
        if (a.C > 0x7fffffffLL) return fail(BINF_E_UNSUPPORTED, "row_sum: too many rows");
        row_sum_block_kernel<OP><<<dim3((unsigned)a.C), 256, 0, st>>>(a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "row_sum launch");
    return 0;
}

}  // namespace binf

using namespace binf;

extern "C" int32_t binf_row_sum_f64(const double *x, double *out, int64_t C,
                                    int64_t D, int32_t op, double shift,
                                    double scale, void *stream)
{
    if (C < 0 || D < 0) return fail(BINF_E_ARG, "row_sum: negative size");
    if (op < 0 || op > 2) return fail(BINF_E_ARG, "row_sum: unknown op %d", op);
    if (C == 0) return 0;
    if ((!x && D > 0) || !out) return fail(BINF_E_ARG, "row_sum: null buffer");
    if (D > 0x7fffffffLL) return fail(BINF_E_UNSUPPORTED, "row_sum: D too large");
    RowSumArgs a;
    a.x = x; a.out = out; a.C = C; a.D = (int32_t)D; a.shift = shift; a.scale = scale;
    a.H = pairwise_tree_height(D < NPY_BUFSIZE ? D : NPY_BUFSIZE);
    hipStream_t st = (hipStream_t)stream;
    switch (op) {
    case OP_SUM: return row_sum_launch<OP_SUM>(a, st);
    case OP_SUMSQ: return row_sum_launch<OP_SUMSQ>(a, st);
    default: return row_sum_launch<OP_SUMSQ_SHIFT>(a, st);
    }
}

static int32_t ew_launch(int kind, double *y, const double *x, double timestep,
                         const double *dt_chain, int32_t half, double k, double x0,
                         int64_t C, int64_t D, int32_t mode, void *stream,
                         const char *what)
{
    if (C < 0 || D < 0) return fail(BINF_E_ARG, "%s: negative size", what);
    if (mode != BINF_MODE_EXACT && mode != BINF_MODE_FMA)
        return fail(BINF_E_ARG, "%s: unknown mode %d", what, mode);
    if (C == 0 || D == 0) return 0;
    if (!y || !x) return fail(BINF_E_ARG, "%s: null buffer", what);
    if (C > 65535) {
        // gridDim.y limit: split the chain range
        const int64_t half_c = C / 2;
        int32_t rc = ew_launch(kind, y, x, timestep, dt_chain, half, k, x0, half_c, D, mode, stream, what);
        if (rc) return rc;
        return ew_launch(kind, y + half_c * D, x + half_c * D, timestep,
                         dt_chain ? dt_chain + half_c : nullptr, half, k, x0,
                         C - half_c, D, mode, stream, what);
    }
    EwArgs a;
    a.y = y; a.x = x; a.dt_chain = dt_chain; a.timestep = timestep; a.k = k;
    a.x0 = x0; a.C = C; a.D = D; a.half = half;
    int64_t bx = (D + 255) / 256;
    if (bx > 64) bx = 64;
    dim3 grid((unsigned)bx, (unsigned)C);
    hipStream_t st = (hipStream_t)stream;
    const bool fma = mode == BINF_MODE_FMA;
    if (kind == EW_KICK) {
        if (fma) ew_kernel<EW_KICK, true><<<grid, 256, 0, st>>>(a);
        else     ew_kernel<EW_KICK, false><<<grid, 256, 0, st>>>(a);
    } else if (kind == EW_DRIFT) {
        if (fma) ew_kernel<EW_DRIFT, true><<<grid, 256, 0, st>>>(a);
        else     ew_kernel<EW_DRIFT, false><<<grid, 256, 0, st>>>(a);
    } else {
        ew_kernel<EW_GAUSS_GRAD, false><<<grid, 256, 0, st>>>(a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what);
    return 0;
}

extern "C" int32_t binf_leapfrog_kick_f64(double *p, const double *grad,
                                          double timestep, const double *dt_chain,
                                          int32_t half, int64_t C, int64_t D,
                                          int32_t mode, void *stream)
{
    return ew_launch(EW_KICK, p, grad, timestep, dt_chain, half ? 1 : 0, 0.0, 0.0,
                     C, D, mode, stream, "leapfrog_kick");
}

extern "C" int32_t binf_leapfrog_drift_f64(double *q, const double *p,
                                           double timestep, const double *dt_chain,
                                           int64_t C, int64_t D, int32_t mode,
                                           void *stream)
{
    return ew_launch(EW_DRIFT, q, p, timestep, dt_chain, 0, 0.0, 0.0, C, D, mode,
                     stream, "leapfrog_drift");
}

extern "C" int32_t binf_gauss_grad_f64(const double *x, double *out, double k,
                                       double x0, int64_t C, int64_t D,
                                       void *stream)
{
    return ew_launch(EW_GAUSS_GRAD, out, x, 0.0, nullptr, 0, k, x0, C, D,
                     BINF_MODE_EXACT, stream, "gauss_grad");
}

extern "C" int32_t binf_accept_select_f64(const double *q_prop, const double *q_old,
                                          const double *e_before, const double *e_after,
                                          const double *u, double *q_out,
                                          uint8_t *accepted, int64_t *n_accepted,
                                          double *dt_chain,
                                          int32_t adapt, double uprate, double downrate,
                                          int64_t C, int64_t D, void *stream)
{
    if (C < 0 || D < 0) return fail(BINF_E_ARG, "accept_select: negative size");
    if (C == 0) return 0;
    if (!q_prop || !q_old || !e_before || !e_after || !u || !q_out || !accepted)
        return fail(BINF_E_ARG, "accept_select: null buffer");
    if (adapt && !dt_chain) return fail(BINF_E_ARG, "accept_select: adapt needs dt_chain");
    if (C > 0x7fffffffLL) return fail(BINF_E_UNSUPPORTED, "accept_select: too many chains");
    AcceptArgs a;
    a.q_prop = q_prop; a.q_old = q_old; a.e_before = e_before; a.e_after = e_after;
    a.u = u; a.q_out = q_out; a.accepted = accepted; a.n_accepted = n_accepted; a.dt_chain = dt_chain;
    a.uprate = uprate; a.downrate = downrate; a.C = C; a.D = D; a.adapt = adapt;
    accept_select_kernel<<<dim3((unsigned)C), 256, 0, (hipStream_t)stream>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "accept_select launch");
    return 0;
}

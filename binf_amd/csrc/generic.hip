// Generic per-step tier: the pieces of HMCSampler.sample() as separate
// chain-batched kernels, for posteriors whose gradient is evaluated by other
// code (any AbstractBinfPDF-shaped plug-in).  gfx950, wave64.
//
// Reference lines replaced: binf/samplers/hmc.py:116-123 (kick / drift),
// :148,150 (energy reductions), :151-164 (accept, adapt, select) and the
// TestHO gradient binf/pdf/__init__.py:191.
#include "rowsum.hpp"
#include "gauss_common.hpp"

namespace binf {

// ---------------------------------------------------------------------------
// row reductions in numpy's pairwise order (machinery: rowsum.hpp)
// ---------------------------------------------------------------------------
enum { OP_SUM = 0, OP_SUMSQ = 1, OP_SUMSQ_SHIFT = 2, OP_SUMSQ_DIFF = 3, OP_SUMSQ_DIFF_DIV = 4 };

struct RedArgs {
    const double *x;
    const double *y;     // OP_SUMSQ_DIFF*: per-column vector subtracted first
    const double *w;     // OP_SUMSQ_DIFF_DIV: per-column divisor of the square
    double shift;
    int64_t D;
};

template <int OP>
struct RedElem {
    const double *row;
    const double *y;
    const double *w;
    double shift;
    __device__ inline double operator()(int i) const
    {
        const double x = row[i];
        if (OP == OP_SUM) return x;
        if (OP == OP_SUMSQ) return x * x;
        const double d = (OP == OP_SUMSQ_SHIFT) ? x - shift : x - y[i];
        if (OP == OP_SUMSQ_DIFF_DIV) return d * d / w[i];
        return d * d;
    }
};

template <int OP>
struct RedMake {
    __device__ static inline RedElem<OP> make(const RedArgs &a, int64_t row)
    {
        return RedElem<OP>{a.x + row * a.D, a.y, a.w, a.shift};
    }
};

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

// One interior leapfrog step after its gradient call, hmc.py:120 followed by the
// next iteration's :119 (or the closing :122): p -= dt * grad; q += p * dt.
// The same two roundings per element as kick then drift, one pass over memory.
template <bool FMA>
__global__ void __launch_bounds__(256)
kick_drift_kernel(double *q, double *p, const double *g, double timestep,
                  const double *dt_chain, int64_t D)
{
    const int64_t c = blockIdx.y;
    const double dt = dt_chain ? dt_chain[c] : timestep;
    double *qc = q + c * D, *pc = p + c * D;
    const double *gc = g + c * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < D;
         i += (int64_t)gridDim.x * 256) {
        const double pn = FMA ? __builtin_fma(-dt, gc[i], pc[i]) : pc[i] - dt * gc[i];
        pc[i] = pn;
        qc[i] = FMA ? __builtin_fma(pn, dt, qc[i]) : qc[i] + pn * dt;
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
    const bool acc = a.u[c] < exp_clipped_range(x);
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

// csb.numeric.exp: exp(clip(x, -308, 709))
__global__ void __launch_bounds__(256)
clipped_exp_kernel(const double *x, double *out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * 256) {
        double v = x[i];
        v = (v < -308.0) ? -308.0 : v;
        v = (v > 709.0) ? 709.0 : v;
        out[i] = exp_clipped_range(v);
    }
}

}  // namespace binf

using namespace binf;

extern "C" int32_t binf_clipped_exp_f64(const double *x, double *out, int64_t n,
                                        void *stream)
{
    if (n < 0) return fail(BINF_E_ARG, "clipped_exp: negative size");
    if (n == 0) return 0;
    if (!x || !out) return fail(BINF_E_ARG, "clipped_exp: null buffer");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    clipped_exp_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(x, out, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "clipped_exp launch");
    return 0;
}

extern "C" int32_t binf_row_sum_f64(const double *x, double *out, int64_t C,
                                    int64_t D, int32_t op, double shift,
                                    double scale, void *stream)
{
    if (C < 0 || D < 0) return fail(BINF_E_ARG, "row_sum: negative size");
    if (op < 0 || op > 2) return fail(BINF_E_ARG, "row_sum: unknown op %d", op);
    if (C == 0) return 0;
    if ((!x && D > 0) || !out) return fail(BINF_E_ARG, "row_sum: null buffer");
    RedArgs a;
    a.x = x; a.y = nullptr; a.w = nullptr; a.shift = shift; a.D = D;
    hipStream_t st = (hipStream_t)stream;
    switch (op) {
    case OP_SUM: return row_reduce_launch<RedMake<OP_SUM>, RedArgs>(a, C, D, scale, out, st, false, "row_sum");
    case OP_SUMSQ: return row_reduce_launch<RedMake<OP_SUMSQ>, RedArgs>(a, C, D, scale, out, st, false, "row_sum");
    default: return row_reduce_launch<RedMake<OP_SUMSQ_SHIFT>, RedArgs>(a, C, D, scale, out, st, false, "row_sum");
    }
}

extern "C" int32_t binf_row_sumsq_diff_f64(const double *x, const double *y,
                                           const double *w, double *out,
                                           int64_t C, int64_t D, double scale,
                                           void *stream)
{
    if (C < 0 || D < 0) return fail(BINF_E_ARG, "row_sumsq_diff: negative size");
    if (C == 0) return 0;
    if (((!x || !y) && D > 0) || !out) return fail(BINF_E_ARG, "row_sumsq_diff: null buffer");
    RedArgs a;
    a.x = x; a.y = y; a.w = w; a.shift = 0.0; a.D = D;
    if (w)
        return row_reduce_launch<RedMake<OP_SUMSQ_DIFF_DIV>, RedArgs>(a, C, D, scale, out, (hipStream_t)stream, false, "row_sumsq_diff");
    return row_reduce_launch<RedMake<OP_SUMSQ_DIFF>, RedArgs>(a, C, D, scale, out, (hipStream_t)stream, false, "row_sumsq_diff");
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

extern "C" int32_t binf_leapfrog_kick_drift_f64(double *q, double *p, const double *grad,
                                                double timestep, const double *dt_chain,
                                                int64_t C, int64_t D, int32_t mode,
                                                void *stream)
{
    if (C < 0 || D < 0) return fail(BINF_E_ARG, "leapfrog_kick_drift: negative size");
    if (mode != BINF_MODE_EXACT && mode != BINF_MODE_FMA)
        return fail(BINF_E_ARG, "leapfrog_kick_drift: unknown mode %d", mode);
    if (C == 0 || D == 0) return 0;
    if (!q || !p || !grad) return fail(BINF_E_ARG, "leapfrog_kick_drift: null buffer");
    hipStream_t st = (hipStream_t)stream;
    int64_t bx = (D + 255) / 256;
    if (bx > 64) bx = 64;
    for (int64_t c0 = 0; c0 < C; c0 += 65535) {            // gridDim.y limit
        const int64_t cn = (C - c0 < 65535) ? C - c0 : 65535;
        const dim3 grid((unsigned)bx, (unsigned)cn);
        const double *dtc = dt_chain ? dt_chain + c0 : nullptr;
        if (mode == BINF_MODE_FMA)
            kick_drift_kernel<true><<<grid, 256, 0, st>>>(q + c0 * D, p + c0 * D, grad + c0 * D,
                                                         timestep, dtc, D);
        else
            kick_drift_kernel<false><<<grid, 256, 0, st>>>(q + c0 * D, p + c0 * D, grad + c0 * D,
                                                          timestep, dtc, D);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "leapfrog_kick_drift");
    return 0;
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

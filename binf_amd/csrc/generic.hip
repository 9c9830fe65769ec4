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
//
// Flat grid over the C*D elements, 16-byte accesses: a thread owns UNR aligned
// pairs of doubles (the pairs of one unroll step are contiguous across the
// workgroup, so every load / store instruction of a wave covers 1 KiB).  With an
// even D both elements of a pair belong to one chain; odd D or a pointer that is
// only 8-byte aligned (a view into a larger tensor) takes the VEC = 1 variant.
// The per-chain step size, if any, is looked up from the element index.  The
// arithmetic per element is unchanged, so the bits are those of the reference's
// elementwise numpy expressions whatever the launch shape.
// ---------------------------------------------------------------------------
struct EwArgs {
    double *y;
    const double *x;
    const double *dt_chain;
    double timestep;
    double k;
    double x0;
    int64_t n;          // C * D
    int64_t D;
    int32_t half;
};

enum { EW_KICK = 0, EW_DRIFT = 1, EW_GAUSS_GRAD = 2 };
constexpr int EW_UNR = 2;

// chain of element i (row-major [C x D]); 32-bit division whenever it fits
__device__ inline int64_t chain_of(int64_t i, int64_t D, bool small)
{
    return small ? (int64_t)((uint32_t)i / (uint32_t)D) : i / D;
}


template <int VEC>
__device__ inline void ew_load(double (&r)[VEC], const double *p, int64_t i)
{
    if (VEC == 2) {
        const double2 v = *reinterpret_cast<const double2 *>(p + i);
        r[0] = v.x;
        r[VEC - 1] = v.y;
    } else {
        r[0] = p[i];
    }
}

template <int VEC>
__device__ inline void ew_store(double *p, int64_t i, const double (&r)[VEC])
{
    if (VEC == 2) {
        double2 v;
        v.x = r[0];
        v.y = r[VEC - 1];
        *reinterpret_cast<double2 *>(p + i) = v;
    } else {
        p[i] = r[0];
    }
}

template <int KIND, bool FMA, int VEC>
__global__ void __launch_bounds__(256) ew_kernel(const EwArgs a)
{
    const bool small = a.n <= 0xffffffffLL;
    const int64_t span = (int64_t)256 * VEC * EW_UNR;            // elements per workgroup pass
    for (int64_t b0 = (int64_t)blockIdx.x * span; b0 < a.n; b0 += (int64_t)gridDim.x * span) {
        double xv[EW_UNR][VEC], yv[EW_UNR][VEC], dt[EW_UNR];
        int64_t idx[EW_UNR];
#pragma unroll
        for (int r = 0; r < EW_UNR; ++r) {
            idx[r] = b0 + ((int64_t)r * 256 + threadIdx.x) * VEC;
            if (idx[r] < a.n) {
                ew_load<VEC>(xv[r], a.x, idx[r]);
                if (KIND != EW_GAUSS_GRAD) ew_load<VEC>(yv[r], a.y, idx[r]);
                dt[r] = a.dt_chain ? a.dt_chain[chain_of(idx[r], a.D, small)] : a.timestep;
            }
        }
#pragma unroll
        for (int r = 0; r < EW_UNR; ++r) {
            if (idx[r] >= a.n) continue;
            double d = dt[r];
            if (KIND == EW_KICK && a.half) d = 0.5 * d;           // "0.5 * timestep" first
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                if (KIND == EW_KICK)            // p -= dt * grad      hmc.py:116,120,123
                    yv[r][v] = FMA ? __builtin_fma(-d, xv[r][v], yv[r][v]) : yv[r][v] - d * xv[r][v];
                else if (KIND == EW_DRIFT)      // q += p * dt         hmc.py:119,122
                    yv[r][v] = FMA ? __builtin_fma(xv[r][v], d, yv[r][v]) : yv[r][v] + xv[r][v] * d;
                else                            // k * (x - x0)        pdf/__init__.py:191
                    yv[r][v] = a.k * (xv[r][v] - a.x0);
            }
            ew_store<VEC>(a.y, idx[r], yv[r]);
        }
    }
}

// One interior leapfrog step after its gradient call, hmc.py:120 followed by the
// next iteration's :119 (or the closing :122): p -= dt * grad; q += p * dt.
// The same two roundings per element as kick then drift, one pass over memory
// (40 bytes per element).
template <bool FMA, int VEC>
__global__ void __launch_bounds__(256)
kick_drift_kernel(double *q, double *p, const double *g, double timestep,
                  const double *dt_chain, int64_t n, int64_t D)
{
    const bool small = n <= 0xffffffffLL;
    const int64_t span = (int64_t)256 * VEC * EW_UNR;
    for (int64_t b0 = (int64_t)blockIdx.x * span; b0 < n; b0 += (int64_t)gridDim.x * span) {
        double qv[EW_UNR][VEC], pv[EW_UNR][VEC], gv[EW_UNR][VEC], dt[EW_UNR];
        int64_t idx[EW_UNR];
#pragma unroll
        for (int r = 0; r < EW_UNR; ++r) {
            idx[r] = b0 + ((int64_t)r * 256 + threadIdx.x) * VEC;
            if (idx[r] < n) {
                ew_load<VEC>(gv[r], g, idx[r]);
                ew_load<VEC>(pv[r], p, idx[r]);
                ew_load<VEC>(qv[r], q, idx[r]);
                dt[r] = dt_chain ? dt_chain[chain_of(idx[r], D, small)] : timestep;
            }
        }
#pragma unroll
        for (int r = 0; r < EW_UNR; ++r) {
            if (idx[r] >= n) continue;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const double pn = FMA ? __builtin_fma(-dt[r], gv[r][v], pv[r][v])
                                      : pv[r][v] - dt[r] * gv[r][v];
                pv[r][v] = pn;
                qv[r][v] = FMA ? __builtin_fma(pn, dt[r], qv[r][v]) : qv[r][v] + pn * dt[r];
            }
            ew_store<VEC>(p, idx[r], pv[r]);
            ew_store<VEC>(q, idx[r], qv[r]);
        }
    }
}

// 16-byte accesses need an even row length and 16-byte aligned bases
static inline bool ew_can_vec2(int64_t D, const void *a, const void *b, const void *c = nullptr)
{
    return (D % 2 == 0) && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
}

static inline unsigned ew_blocks(int64_t n, int vec)
{
    const int64_t span = (int64_t)256 * vec * EW_UNR;
    int64_t b = (n + span - 1) / span;
    if (b > (1 << 20)) b = 1 << 20;                  // grid-stride beyond 2^20 workgroups
    return (unsigned)b;
}

// ---------------------------------------------------------------------------
// Metropolis accept, step-size adaption, select           hmc.py:151-164,188-191
//
// A chain is served by LPC = 8 .. 256 threads of a workgroup (a power of two
// sized to the row), so short rows (the polynomial model's K = 33 coefficients)
// share a workgroup; every thread of the group evaluates the same accept test
// (same inputs, same bits) and copies its part of the selected row with 16-byte
// accesses where alignment allows.
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
    int32_t lpc;        // threads per chain (power of two, 8..256)
};

template <int VEC>
__global__ void __launch_bounds__(256) accept_select_kernel(const AcceptArgs a)
{
    const int lpc = a.lpc;
    const int sub = threadIdx.x & (lpc - 1);
    const int64_t c = (int64_t)blockIdx.x * (256 / lpc) + threadIdx.x / lpc;
    if (c >= a.C) return;
    double x = -(a.e_after[c] - a.e_before[c]);
    x = (x < -308.0) ? -308.0 : x;
    x = (x > 709.0) ? 709.0 : x;
    const bool acc = a.u[c] < exp_clipped_range(x);
    const double *src = (acc ? a.q_prop : a.q_old) + c * a.D;
    double *dst = a.q_out + c * a.D;
    if (dst != src) {
        for (int64_t i = (int64_t)sub * VEC; i < a.D; i += (int64_t)lpc * VEC) {
            double r[VEC];
            ew_load<VEC>(r, src, i);
            ew_store<VEC>(dst, i, r);
        }
    }
    if (sub == 0) {
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

// E[c] = -log_prob[c] + 0.5 * np.sum(p[c]**2) (hmc.py:143,148,150) in one launch:
// the kinetic row sum with the subtraction as its epilogue.  (-lp) + k and k - lp
// are the same IEEE operation, so the bits equal the three-launch form.
extern "C" int32_t binf_hmc_energy_f64(const double *p, const double *log_prob, double *out,
                                       int64_t C, int64_t D, void *stream)
{
    if (C < 0 || D < 0) return fail(BINF_E_ARG, "hmc_energy: negative size");
    if (C == 0) return 0;
    if ((!p && D > 0) || !log_prob || !out) return fail(BINF_E_ARG, "hmc_energy: null buffer");
    RedArgs a;
    a.x = p; a.y = nullptr; a.w = nullptr; a.shift = 0.0; a.D = D;
    GaussFinish fin;
    fin.on = 2; fin.tau = 1.0; fin.tau_chain = nullptr; fin.n_data = 0.0; fin.minus = log_prob;
    return row_reduce_launch<RedMake<OP_SUMSQ>, RedArgs>(a, C, D, 0.5, out, (hipStream_t)stream, false,
                                                        "hmc_energy", 0, false, &fin);
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

template <int VEC>
static void ew_dispatch(int kind, bool fma, const EwArgs &a, hipStream_t st)
{
    const dim3 grid(ew_blocks(a.n, VEC));
    if (kind == EW_KICK) {
        if (fma) ew_kernel<EW_KICK, true, VEC><<<grid, 256, 0, st>>>(a);
        else     ew_kernel<EW_KICK, false, VEC><<<grid, 256, 0, st>>>(a);
    } else if (kind == EW_DRIFT) {
        if (fma) ew_kernel<EW_DRIFT, true, VEC><<<grid, 256, 0, st>>>(a);
        else     ew_kernel<EW_DRIFT, false, VEC><<<grid, 256, 0, st>>>(a);
    } else {
        ew_kernel<EW_GAUSS_GRAD, false, VEC><<<grid, 256, 0, st>>>(a);
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
    if (C > 0x7fffffffffffffffLL / D) return fail(BINF_E_ARG, "%s: C*D overflows", what);
    EwArgs a;
    a.y = y; a.x = x; a.dt_chain = dt_chain; a.timestep = timestep; a.k = k;
    a.x0 = x0; a.n = C * D; a.D = D; a.half = half;
    hipStream_t st = (hipStream_t)stream;
    const bool fma = mode == BINF_MODE_FMA;
    if (ew_can_vec2(D, y, x)) ew_dispatch<2>(kind, fma, a, st);
    else                      ew_dispatch<1>(kind, fma, a, st);
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
    if (C > 0x7fffffffffffffffLL / D) return fail(BINF_E_ARG, "leapfrog_kick_drift: C*D overflows");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = C * D;
    const bool fma = mode == BINF_MODE_FMA;
    if (ew_can_vec2(D, q, p, grad)) {
        const dim3 grid(ew_blocks(n, 2));
        if (fma) kick_drift_kernel<true, 2><<<grid, 256, 0, st>>>(q, p, grad, timestep, dt_chain, n, D);
        else     kick_drift_kernel<false, 2><<<grid, 256, 0, st>>>(q, p, grad, timestep, dt_chain, n, D);
    } else {
        const dim3 grid(ew_blocks(n, 1));
        if (fma) kick_drift_kernel<true, 1><<<grid, 256, 0, st>>>(q, p, grad, timestep, dt_chain, n, D);
        else     kick_drift_kernel<false, 1><<<grid, 256, 0, st>>>(q, p, grad, timestep, dt_chain, n, D);
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
    AcceptArgs a;
    a.q_prop = q_prop; a.q_old = q_old; a.e_before = e_before; a.e_after = e_after;
    a.u = u; a.q_out = q_out; a.accepted = accepted; a.n_accepted = n_accepted; a.dt_chain = dt_chain;
    a.uprate = uprate; a.downrate = downrate; a.C = C; a.D = D; a.adapt = adapt;
    const bool vec2 = ew_can_vec2(D, q_prop, q_old, q_out);
    const int64_t per_thread = vec2 ? 2 : 1;
    int lpc = 8;
    while (lpc < 256 && (int64_t)lpc * per_thread < D) lpc <<= 1;
    a.lpc = lpc;
    const int64_t blocks = (C + 256 / lpc - 1) / (256 / lpc);
    if (blocks > 0x7fffffffLL) return fail(BINF_E_UNSUPPORTED, "accept_select: too many chains");
    if (vec2) accept_select_kernel<2><<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(a);
    else      accept_select_kernel<1><<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "accept_select launch");
    return 0;
}

// Posterior-predictive density of a Gaussian error model over a grid of (x, y) points:
// what binf/example/misc.py:3-16 (predict) computes for one point with a Python loop over
// the samples, and binf/example/plots.py:8-11 (plot_prediction_tube) repeats for every
// point of a [n_x x n_ys] grid -- here ONE launch for the whole grid, the consumer side of
// the sample store.
//
//   out[i, j] = exp(log_sum_exp_s f[s, i, j]) / S
//   f[s, i, j] = -0.5 * (mock[s, i] - ys[i, j])**2 * precision[s]
//                + 0.5 * log(precision[s]) - 0.5 * log(2 pi)                 (misc.py:8)
//   log_sum_exp(x) = log(sum(exp(x - max(x)))) + max(x)    (csb.numeric.log_sum_exp; csb is
//                                                           absent: its published definition)
//
// mock[s, i] is the forward model at predict_space[i] for sample s -- for the polynomial
// model binf_poly_forward_f64's output, for a user's forward model whatever it returns.
//
// Layout: a workgroup owns 16 x-points and JT y-values per point; its 256 threads are
// 16 (x, fastest: mock rows are read 128 B at a time) x 16 (sample lanes).  Two passes over
// the samples (max, then sum of exp), each joined across the 16 sample lanes through LDS.
// A small grid with many samples (the example: 100 x 150 points, 51200 samples) would fill
// half the chip with 133 workgroups: the samples are then cut into chunks (blockIdx.z), each
// chunk leaves its (max, sum of exp) pair in a workspace and a second launch joins them --
// log_sum_exp over chunks, sum_z s_z exp(M_z - M); chunks whose terms are all -inf drop out.
// NaN handling follows numpy: a NaN term (precision < 0) makes max, and so the density, NaN;
// precision == 0 gives -inf terms, all -inf gives NaN (inf - inf), as the reference would.
// The sum runs in lane-strided order, not numpy's pairwise order, and log / exp are the
// device library's: within ~1e-14 of the numpy restatement, not bit-identical
// (tests/test_gpu_predict.py states the tolerance).
#include "common.hpp"
#include "gauss_common.hpp"

namespace binf {

constexpr int PRED_XT = 16;      // x-points per workgroup
constexpr int PRED_SL = 16;      // sample lanes per x-point
constexpr int PRED_JT = 8;       // y-values per x-point and workgroup (4096 samples x 1000 x 1000
                                 // points: 7.6 ms with 8, 10.4 with 4, 15.9 with 2 -- log(precision)
                                 // and the mock load are shared by the JT terms; scripts/probe_predict.py)

struct PredictArgs {
    const double *mock;          // [S x nx]
    const double *precision;     // [S]
    const double *ys;            // [nx x ny]
    double *out;                 // [nx x ny]
    int64_t S, nx, ny;
    double half_log_2pi;
    double *part;                // [nsplit x nx x ny x 2] (max, sum) per sample chunk, or null
    int64_t chunk;               // samples per chunk (blockIdx.z walks the chunks)
};

// max that keeps a NaN once it has seen one (np.max propagates NaN)
__device__ inline double nan_max(double m, double f) { return (f > m || f != f) ? f : m; }

__device__ inline double predictive_term(double m, double y, double prec, double hlp,
                                         double half_log_2pi)
{
    const double d = m - y;
    return (-0.5 * (d * d)) * prec + hlp - half_log_2pi;
}

__global__ void __launch_bounds__(PRED_XT * PRED_SL)
predictive_density_kernel(const PredictArgs a)
{
    __shared__ double red[PRED_JT][PRED_SL][PRED_XT + 1];
    const int tx = threadIdx.x % PRED_XT, sl = threadIdx.x / PRED_XT;
    const int64_t i = (int64_t)blockIdx.x * PRED_XT + tx;
    const int64_t j0 = (int64_t)blockIdx.y * PRED_JT;
    const bool live = i < a.nx;
    const int64_t ii = live ? i : a.nx - 1;

    double y[PRED_JT], M[PRED_JT], sum[PRED_JT];
#pragma unroll
    for (int j = 0; j < PRED_JT; ++j) {
        const int64_t jj = (j0 + j < a.ny) ? j0 + j : a.ny - 1;
        y[j] = a.ys[ii * a.ny + jj];
        M[j] = -__builtin_inf();
        sum[j] = 0.0;
    }
    const int64_t s_lo = (int64_t)blockIdx.z * a.chunk;
    const int64_t s_hi = (s_lo + a.chunk < a.S) ? s_lo + a.chunk : a.S;
    // pass 1: the maximum over the samples
    for (int64_t s = s_lo + sl; s < s_hi; s += PRED_SL) {
        const double m = a.mock[s * a.nx + ii], prec = a.precision[s];
        const double hlp = 0.5 * log(prec);
#pragma unroll
        for (int j = 0; j < PRED_JT; ++j)
            M[j] = nan_max(M[j], predictive_term(m, y[j], prec, hlp, a.half_log_2pi));
    }
#pragma unroll
    for (int j = 0; j < PRED_JT; ++j) red[j][sl][tx] = M[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PRED_JT; ++j) {
        double m = red[j][0][tx];
        for (int l = 1; l < PRED_SL; ++l) m = nan_max(m, red[j][l][tx]);
        M[j] = m;
    }
    __syncthreads();
    // pass 2: sum of exp(f - max)
    for (int64_t s = s_lo + sl; s < s_hi; s += PRED_SL) {
        const double m = a.mock[s * a.nx + ii], prec = a.precision[s];
        const double hlp = 0.5 * log(prec);
#pragma unroll
        for (int j = 0; j < PRED_JT; ++j)
            sum[j] += np_exp(predictive_term(m, y[j], prec, hlp, a.half_log_2pi) - M[j]);
    }
#pragma unroll
    for (int j = 0; j < PRED_JT; ++j) red[j][sl][tx] = sum[j];
    __syncthreads();
    if (sl == 0 && live) {
#pragma unroll
        for (int j = 0; j < PRED_JT; ++j) {
            if (j0 + j >= a.ny) break;
            double t = red[j][0][tx];
            for (int l = 1; l < PRED_SL; ++l) t += red[j][l][tx];
            if (a.part) {                                           // one chunk of the samples
                double *o = a.part + (((int64_t)blockIdx.z * a.nx + i) * a.ny + j0 + j) * 2;
                o[0] = M[j];
                o[1] = t;
                continue;
            }
            const double lse = log(t) + M[j];                       // log_sum_exp
            a.out[i * a.ny + j0 + j] = np_exp(lse) / (double)a.S;   // misc.py:16
        }
    }
}

// log_sum_exp over the chunks' (max, sum) pairs: M = max_z M_z, sum_z s_z exp(M_z - M).  A chunk
// whose terms were all -inf (M_z = -inf, s_z = NaN from inf - inf) contributes nothing unless
// every chunk is like that -- then the result is NaN, as numpy's on the whole array.
__global__ void predictive_join_kernel(const double *part, double *out, int64_t npts, int nsplit,
                                       double S)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= npts) return;
    double M = -__builtin_inf();
    for (int z = 0; z < nsplit; ++z) M = nan_max(M, part[((int64_t)z * npts + g) * 2]);
    double t = 0.0;
    if (M == -__builtin_inf()) {
        t = __builtin_nan("");
    } else {
        for (int z = 0; z < nsplit; ++z) {
            const double mz = part[((int64_t)z * npts + g) * 2], sz = part[((int64_t)z * npts + g) * 2 + 1];
            if (mz == -__builtin_inf()) continue;
            t += sz * np_exp(mz - M);
        }
    }
    out[g] = np_exp(log(t) + M) / S;
}

}  // namespace binf

using namespace binf;

// Sample chunks for a grid of bx x by workgroups: enough workgroups to fill the chip a few
// times over, at least 256 samples per chunk.
static int predict_splits(int64_t S, int64_t bx, int64_t by)
{
    const int64_t blocks = bx * by;
    int64_t ns = (1024 + blocks - 1) / blocks;
    if (ns > S / 256) ns = S / 256;
    if (ns > 64) ns = 64;
    return ns < 2 ? 1 : (int)ns;
}

extern "C" int64_t binf_predictive_density_workspace_bytes(int64_t S, int64_t nx, int64_t ny)
{
    if (S < 1 || nx < 1 || ny < 1) return 0;
    const int ns = predict_splits(S, (nx + PRED_XT - 1) / PRED_XT, (ny + PRED_JT - 1) / PRED_JT);
    return ns == 1 ? 0 : (int64_t)ns * nx * ny * 2 * (int64_t)sizeof(double);
}

extern "C" int32_t binf_predictive_density_f64(const double *mock, const double *precision,
                                               const double *ys, double *out, int64_t S,
                                               int64_t nx, int64_t ny, double half_log_2pi,
                                               void *workspace, int64_t workspace_bytes,
                                               void *stream)
{
    if (S < 1 || nx < 0 || ny < 0)
        return fail(BINF_E_ARG, "predictive_density: need S>=1 (max of no samples is undefined), nx>=0, ny>=0");
    if (nx == 0 || ny == 0) return 0;
    if (!mock || !precision || !ys || !out)
        return fail(BINF_E_ARG, "predictive_density: null buffer");
    const int64_t bx = (nx + PRED_XT - 1) / PRED_XT, by = (ny + PRED_JT - 1) / PRED_JT;
    if (bx > 0x7fffffffLL || by > 65535)
        return fail(BINF_E_UNSUPPORTED, "predictive_density: grid of %lld x %lld points too large "
                    "(ny <= %d per call)", (long long)nx, (long long)ny, 65535 * PRED_JT);
    if (overlap_f64(out, nx * ny, mock, S * nx) || overlap_f64(out, nx * ny, precision, S) ||
        overlap_f64(out, nx * ny, ys, nx * ny))
        return fail(BINF_E_ALIAS, "predictive_density: out overlaps an input");
    const int ns = predict_splits(S, bx, by);
    const int64_t need = binf_predictive_density_workspace_bytes(S, nx, ny);
    if (need > 0 && (!workspace || workspace_bytes < need))
        return fail(BINF_E_ARG, "predictive_density: needs %lld bytes of workspace "
                    "(binf_predictive_density_workspace_bytes), got %lld",
                    (long long)need, (long long)workspace_bytes);
    if (need > 0 && (overlap_f64(workspace, need / 8, out, nx * ny) || overlap_f64(workspace, need / 8, mock, S * nx) ||
                     overlap_f64(workspace, need / 8, ys, nx * ny) || overlap_f64(workspace, need / 8, precision, S)))
        return fail(BINF_E_ALIAS, "predictive_density: the workspace overlaps a buffer");
    hipStream_t st = (hipStream_t)stream;
    PredictArgs a;
    a.mock = mock; a.precision = precision; a.ys = ys; a.out = out;
    a.S = S; a.nx = nx; a.ny = ny; a.half_log_2pi = half_log_2pi;
    a.part = ns > 1 ? (double *)workspace : nullptr;
    a.chunk = ns > 1 ? (S + ns - 1) / ns : S;
    predictive_density_kernel<<<dim3((unsigned)bx, (unsigned)by, (unsigned)ns), PRED_XT * PRED_SL, 0, st>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "predictive_density launch");
    if (ns > 1) {
        const int64_t npts = nx * ny;
        predictive_join_kernel<<<dim3((unsigned)((npts + 255) / 256)), 256, 0, st>>>(
            (const double *)workspace, out, npts, ns, (double)S);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "predictive_density join launch");
    }
    return 0;
}

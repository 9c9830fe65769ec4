// Pairwise-distance-restraint model (BASELINE config C5, "chromatin-like"):
// n beads in 3-D per chain, forward model = all n(n-1)/2 pair distances,
// Gaussian error model on the distances.  gfx950, wave64.
//
// No reference code exists for this model (reference README.rst:9 only names
// the application); it is build-defined in the shape of the reference's
// plug-in surface (AbstractForwardModel / AbstractErrorModel,
// binf/model/forwardmodels.py:10-66, binf/pdf/likelihoods.py:141-155).
//
//   forward : d_p = sqrt(((xi-xj)**2).sum()) for pair p = (i<j) in np.triu order
//   energy gradient w.r.t. bead i:  tau * sum_{j != i} (d_ij - y_ij) (x_i - x_j)/d_ij
//
// The force kernel is the O(n^2) all-pairs loop, one workgroup per chain with
// the chain's coordinates staged in LDS; the [3n x n(n-1)/2] Jacobian the
// generic Likelihood path would need (200 MB per chain at n = 256) is never
// formed.
#include "common.hpp"

namespace binf {

// d[c, p] for p-th pair (I[p], J[p]); coordinates x[c, 3*bead + axis]
__global__ void __launch_bounds__(256)
pairdist_forward_kernel(const double *x, const int32_t *I, const int32_t *J,
                        double *out, int64_t n_beads, int64_t n_pairs)
{
    const int64_t c = blockIdx.y;
    const double *xc = x + c * 3 * n_beads;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_pairs;
         p += (int64_t)gridDim.x * 256) {
        const int i = I[p], j = J[p];
        const double a = xc[3 * i] - xc[3 * j];
        const double b = xc[3 * i + 1] - xc[3 * j + 1];
        const double e = xc[3 * i + 2] - xc[3 * j + 2];
        // np.sum(diff**2, axis=1): sequential over the 3 components
        const double s = (a * a + b * b) + e * e;
        out[c * n_pairs + p] = sqrt(s);
    }
}

// out[c, 3i + a] = tau_c * sum_{j != i} (d_ij - y[j][i]) * (x_i - x_j)[a] / d_ij
// ymat: symmetric [n x n] target distances (diagonal ignored).
template <int TILE>
__global__ void __launch_bounds__(256)
pairdist_grad_kernel(const double *x, const double *ymat, double tau,
                     const double *tau_chain, double *out, int32_t n_beads)
{
    __shared__ double sx[TILE][3];
    const int64_t c = blockIdx.x;
    const double *xc = x + c * 3 * (int64_t)n_beads;
    const double t = tau_chain ? tau_chain[c] : tau;
    for (int i0 = 0; i0 < n_beads; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const bool iv = i < n_beads;
        double xi0 = 0, xi1 = 0, xi2 = 0;
        if (iv) { xi0 = xc[3 * i]; xi1 = xc[3 * i + 1]; xi2 = xc[3 * i + 2]; }
        double f0 = 0.0, f1 = 0.0, f2 = 0.0;
        for (int j0 = 0; j0 < n_beads; j0 += TILE) {
            __syncthreads();
            for (int k = threadIdx.x; k < TILE * 3; k += 256) {
                const int idx = 3 * j0 + k;
                (&sx[0][0])[k] = (idx < 3 * n_beads) ? xc[idx] : 0.0;
            }
            __syncthreads();
            const int jn = (n_beads - j0 < TILE) ? n_beads - j0 : TILE;
            if (iv) {
                for (int jj = 0; jj < jn; ++jj) {
                    const int j = j0 + jj;
                    if (j == i) continue;
                    const double a = xi0 - sx[jj][0];
                    const double b = xi1 - sx[jj][1];
                    const double e = xi2 - sx[jj][2];
                    const double d = sqrt((a * a + b * b) + e * e);
                    const double w = (d - ymat[(int64_t)j * n_beads + i]) / d;
                    f0 += w * a;
                    f1 += w * b;
                    f2 += w * e;
                }
            }
        }
        if (iv) {
            double *o = out + c * 3 * (int64_t)n_beads + 3 * i;
            o[0] = t * f0;
            o[1] = t * f1;
            o[2] = t * f2;
        }
    }
}

}  // namespace binf

using namespace binf;

extern "C" int32_t binf_pairdist_forward_f64(const double *x, const int32_t *pair_i,
                                             const int32_t *pair_j, double *out,
                                             int64_t C, int64_t n_beads,
                                             int64_t n_pairs, void *stream)
{
    if (C < 0 || n_beads < 1 || n_pairs < 0)
        return fail(BINF_E_ARG, "pairdist_forward: bad sizes");
    if (C == 0 || n_pairs == 0) return 0;
    if (!x || !pair_i || !pair_j || !out)
        return fail(BINF_E_ARG, "pairdist_forward: null buffer");
    if (C > 65535) return fail(BINF_E_UNSUPPORTED, "pairdist_forward: more than 65535 chains per call");
    int64_t bx = (n_pairs + 255) / 256;
    if (bx > 1024) bx = 1024;
    pairdist_forward_kernel<<<dim3((unsigned)bx, (unsigned)C), 256, 0, (hipStream_t)stream>>>(
        x, pair_i, pair_j, out, n_beads, n_pairs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "pairdist_forward launch");
    return 0;
}

extern "C" int32_t binf_pairdist_gauss_grad_f64(const double *x, const double *ymat,
                                                double precision,
                                                const double *precision_chain,
                                                double *out, int64_t C,
                                                int64_t n_beads, void *stream)
{
    if (C < 0 || n_beads < 1) return fail(BINF_E_ARG, "pairdist_gauss_grad: bad sizes");
    if (C == 0) return 0;
    if (!x || !ymat || !out) return fail(BINF_E_ARG, "pairdist_gauss_grad: null buffer");
    if (n_beads > 46340 || C > 0x7fffffffLL)
        return fail(BINF_E_UNSUPPORTED, "pairdist_gauss_grad: too large");
    pairdist_grad_kernel<256><<<dim3((unsigned)C), 256, 0, (hipStream_t)stream>>>(
        x, ymat, precision, precision_chain, out, (int32_t)n_beads);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "pairdist_gauss_grad launch");
    return 0;
}

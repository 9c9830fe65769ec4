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
// Force kernels (the [3n x n(n-1)/2] Jacobian the generic Likelihood path would
// need -- 200 MB per chain at n = 256 -- is never formed):
//   32 <= n <= 256  every unordered pair once, target distances in registers, one
//                   workgroup of 1 / 4 / 9 / 16 waves per chain (second half of this file);
//   257 .. 1024     (with packed targets) every unordered pair once as well: the "ring"
//                   scheme -- a wave per block of 64 row beads, block pairs walked in phases,
//                   targets streamed from the packed array (end of the device code);
//   other n         one-sided all-pairs loops, the chain's coordinates staged in LDS,
//                   target distances read from a symmetric [n x n] matrix.
// Each has a force-only kernel and a fused leapfrog (the whole _leapfrog() of
// binf/samplers/hmc.py:92-125 in one launch) that are bit-identical to each other.
//
// Energy side (bit-identical to numpy: correctly rounded sqrt, np.sum's pairwise order):
//   pairdist_chi2_rows_kernel  chi^2 of one or two chains per workgroup, distances formed on
//                              the fly, with the per-chain two-entry memo of rowsum.hpp;
//   pairdist_energy_kernel     HMCSampler's E = 0.5 sum p^2 - log_prob for likelihood + one
//                              isotropic Gaussian prior in one launch (memo check included).
#include <stdlib.h>
#include <atomic>
#include "rowsum.hpp"

namespace binf {

// Correctly rounded sqrt of a squared distance.  The compiler's expansion of sqrt(double)
// is rsq + one coupled Goldschmidt step + two residual corrections, wrapped in a scaling
// of inputs below 2^-767 and a pass-through of 0 / inf (18 instructions, 8 of them that
// wrapping).  For an argument in [2^-767, inf) the wrapping does nothing: the same
// iteration without it gives the same bits; any other argument takes the library path.
__device__ inline double sqrt_rn(double s)
{
    const uint32_t hi = (uint32_t)__double2hiint(s);
    const bool plain = hi - 0x10000000u < 0x7ff00000u - 0x10000000u;
    const double r = __builtin_amdgcn_rsq(s);
    double g = s * r;
    double h = r * 0.5;
    const double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    double d = __builtin_fma(-g, g, s);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, s);
    g = __builtin_fma(d, h, g);
    if (__builtin_expect(!plain, 0)) g = sqrt(s);
    return g;
}

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
        out[c * n_pairs + p] = sqrt_rn(s);
    }
}

// chi^2 of the restraints without the [C x n_pairs] distances ever reaching HBM:
// element p of the row reduction is (d_p - y_p)^2 with d_p computed on the fly,
// exactly as pairdist_forward_kernel + the error model's (mock - ys)**2 do.
struct PairArgs {
    const double *x;
    const int32_t *I;
    const int32_t *J;
    const double *ys;
    int64_t n_beads;
};

struct PairResid {
    const double *xc;
    const int32_t *I;
    const int32_t *J;
    const double *ys;
    __device__ inline double operator()(int p) const
    {
        const int i = I[p], j = J[p];
        const double a = xc[3 * i] - xc[3 * j];
        const double b = xc[3 * i + 1] - xc[3 * j + 1];
        const double e = xc[3 * i + 2] - xc[3 * j + 2];
        const double s = (a * a + b * b) + e * e;
        const double d = sqrt_rn(s) - ys[p];
        return d * d;
    }
};

struct PairResidMake {
    __device__ static inline PairResid make(const PairArgs &a, int64_t row)
    {
        PairResid f;
        f.xc = a.x + row * 3 * a.n_beads;
        f.I = a.I;
        f.J = a.J;
        f.ys = a.ys;
        return f;
    }
    // the chain's coordinates in LDS: 6 of the 9 gathers per pair leave the
    // vector memory pipe, which bounded the kernel (0.22 -> 0.1x ms at 2048 chains)
    __device__ static inline void stage(const PairArgs &a, int64_t row, double *lds)
    {
        const double *xc = a.x + row * 3 * a.n_beads;
        for (int64_t k = threadIdx.x; k < 3 * a.n_beads; k += blockDim.x) lds[k] = xc[k];
    }
    __device__ static inline PairResid make_lds(const PairArgs &a, int64_t row, const double *lds)
    {
        PairResid f;
        f.xc = lds;
        f.I = a.I;
        f.J = a.J;
        f.ys = a.ys;
        return f;
    }
};

// ---- chi^2 of ROWS chains per workgroup: the pair list (I, J, y) is the same for every
// chain, so one pass over it serves ROWS chains -- the index / target loads and their
// address arithmetic are paid once, and each lane has 8 x ROWS independent square roots
// in flight.  With one chain per workgroup the list is streamed from L2 once per chain
// (16 bytes per pair and chain: 1 GB per evaluation at 2048 chains of 256 beads), which
// is what bounded the one-row kernel there.  Same tree, same order as
// row_reduce_block_kernel: the results are bit-identical to the one-row path.
// LDS: coordinates interleaved [3*bead + axis][row], so one bead's ROWS values of an axis
// are one 16- or 32-byte read.
template <int ROWS>
struct PairResidRows {
    const double *xl;
    const int32_t *I;
    const int32_t *J;
    const double *ys;
    __device__ inline void operator()(int p, double (&v)[ROWS]) const
    {
        const int i = 3 * ROWS * I[p], j = 3 * ROWS * J[p];
        const double y = ys[p];
        double xi[3][ROWS], xj[3][ROWS];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                xi[ax][r] = xl[i + ax * ROWS + r];
                xj[ax][r] = xl[j + ax * ROWS + r];
            }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const double a = xi[0][r] - xj[0][r];
            const double b = xi[1][r] - xj[1][r];
            const double e = xi[2][r] - xj[2][r];
            const double s = (a * a + b * b) + e * e;
            const double d = sqrt_rn(s) - y;
            v[r] = d * d;
        }
    }
};

// leaf_sum_f (rowsum.hpp) for ROWS rows at once; U element values per row in flight
template <int ROWS, int U, class F>
__device__ inline void leaf_sum_rows(const F &f, int off, int n, int lane, bool active,
                                     double (&res)[ROWS])
{
    const int j = lane & 7;
    const int T = (n >= 8) ? (n >> 3) : 0;
    const int rem = (n >= 8) ? (n & 7) : n;
    double r[ROWS];
#pragma unroll
    for (int q = 0; q < ROWS; ++q) r[q] = 0.0;
    if (active && T > 0) {
        int t = 0;
        for (; t + U <= T; t += U) {
            double v[U][ROWS];
#pragma unroll
            for (int u = 0; u < U; ++u) f(off + 8 * (t + u) + j, v[u]);
#pragma unroll
            for (int q = 0; q < ROWS; ++q) r[q] = (t == 0) ? v[0][q] : r[q] + v[0][q];
#pragma unroll
            for (int u = 1; u < U; ++u)
#pragma unroll
                for (int q = 0; q < ROWS; ++q) r[q] = r[q] + v[u][q];
        }
        for (; t < T; ++t) {
            double v[ROWS];
            f(off + 8 * t + j, v);
#pragma unroll
            for (int q = 0; q < ROWS; ++q) r[q] = (t == 0) ? v[q] : r[q] + v[q];
        }
    }
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
        const double s8 = sum8_f64(r[q]);
        res[q] = (T > 0) ? s8 : -0.0;
    }
    if (__any(rem != 0)) {
        double tail[ROWS];
#pragma unroll
        for (int q = 0; q < ROWS; ++q) tail[q] = 0.0;
        if (active && j < rem) f(off + 8 * T + j, tail);
        const int leafbase = lane & ~7;
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int q = 0; q < ROWS; ++q) {
                const double v = shfl_f64(tail[q], leafbase + i);
                const double s = res[q] + v;
                res[q] = (i < rem) ? s : res[q];
            }
    }
}

// One np.sum-order reduction of D elements for ROWS rows by the whole workgroup;
// f(i, v[ROWS]) = element i of every row.  THREADS / 8 lane groups sum leaves: with 16
// waves a round covers two 64-leaf chunks at once, with 4 waves a chunk takes two passes.
// The leaf sums of a round go to LDS, ONE barrier, and wave 0 alone walks the round's trees
// with lane exchanges (lane = path) and adds the chunk sums to the running total in chunk
// order -- a tree through LDS costs two barriers per level, 14 per chunk, which at one
// 1024-thread workgroup per CU was a third of the kernel.  The LDS buffers alternate
// between rounds (`round` runs on through consecutive calls), so the barrier of round r + 1
// is all that separates wave 0's reads of round r from the writes of round r + 2.
// The totals come back in THREAD 0 only.
struct BlockSumScratch {
    double S[2][2][128];     // [round parity][row][slot]   (ROWS <= 2)
    int dep[2][128];
};

template <int ROWS, int U, int THREADS, class F>
__device__ inline void block_sum_rows(const F &f, int D, int H, BlockSumScratch &sc, int &round,
                                      double (&total)[ROWS])
{
    static_assert(ROWS <= 2, "BlockSumScratch holds two rows");
    constexpr int GROUPS = THREADS / 8;
    const int npaths = 1 << H;                                   // <= 128
    const int lane = threadIdx.x & 63;
    const int group = threadIdx.x >> 3;
    const int nchunks = D <= 0 ? 1 : (D + NPY_BUFSIZE - 1) / NPY_BUFSIZE;
    const int R = GROUPS >= npaths ? GROUPS / npaths : 1;        // chunks per round
    const int passes = GROUPS >= npaths ? 1 : (npaths + GROUPS - 1) / GROUPS;   // passes per chunk
#pragma unroll
    for (int q = 0; q < ROWS; ++q) total[q] = 0.0;
    for (int c0 = 0; c0 < nchunks; c0 += R, ++round) {
        const int b = round & 1;
        for (int ps = 0; ps < passes; ++ps) {
            const int item = ps * GROUPS + group;                // (chunk of the round, path)
            const int cr = item >> H, path = item & (npaths - 1);
            const int chunk = c0 + cr;
            const bool act = cr < R && chunk < nchunks && item < R * npaths;
            const int cbase = act ? chunk * NPY_BUFSIZE : 0;
            const int n = (D - cbase < NPY_BUFSIZE) ? D - cbase : NPY_BUFSIZE;
            const Leaf L = pairwise_leaf(n, H, act ? path : 0);
            double s[ROWS];
            leaf_sum_rows<ROWS, U>(f, cbase + L.off, L.len, lane, act, s);
            if (act && (lane & 7) == 0) {
#pragma unroll
                for (int q = 0; q < ROWS; ++q) sc.S[b][q][item] = s[q];
                sc.dep[b][item] = L.depth;
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            for (int cr = 0; cr < R && c0 + cr < nchunks; ++cr) {
                const int base = cr << H;
                const bool two = npaths > 64;                    // H = 7: paths p and p + 64 per lane
                const int p0 = base + (lane & (npaths - 1)), p1 = base + ((lane + 64) & (npaths - 1));
                const int d0 = sc.dep[b][p0], d1 = sc.dep[b][p1];
#pragma unroll
                for (int q = 0; q < ROWS; ++q) {
                    double v0 = sc.S[b][q][p0], v1 = sc.S[b][q][p1];
                    for (int l = 0; l < H && l < 6; ++l) {
                        const double o0 = shfl_f64(v0, lane ^ (1 << l));
                        v0 = (d0 >= H - l) ? v0 + o0 : v0;
                        if (two) {
                            const double o1 = shfl_f64(v1, lane ^ (1 << l));
                            v1 = (d1 >= H - l) ? v1 + o1 : v1;
                        }
                    }
                    if (two) v0 = (d0 >= 1) ? v0 + v1 : v0;      // level 6 joins paths p and p + 64
                    total[q] = total[q] + v0;                    // lane 0: path 0, the chunk's sum
                }
            }
        }
    }
}

template <int ROWS, int U, int THREADS>
__global__ void __launch_bounds__(THREADS)
pairdist_chi2_rows_kernel(const PairArgs a, const RowGeom g, double *out)
{
    __shared__ BlockSumScratch sc;
    int round = 0;
    extern __shared__ double row_lds[];
    const int64_t row0 = (int64_t)blockIdx.x * ROWS;
    int64_t rows[ROWS];                          // the last workgroup repeats the last chain
#pragma unroll
    for (int q = 0; q < ROWS; ++q) rows[q] = (row0 + q < g.C) ? row0 + q : g.C - 1;
    if (g.skip) {                                // workgroup-uniform
        bool all_skip = true;
#pragma unroll
        for (int q = 0; q < ROWS; ++q) all_skip = all_skip && g.skip[rows[q]] != 0;
        if (all_skip) {
            if (threadIdx.x < ROWS && row0 + threadIdx.x < g.C) {
                const int64_t row = row0 + threadIdx.x;
                out[row] = row_result(g, row, *row_memo_slot(g, row));
            }
            return;
        }
    }
    const int n3 = 3 * (int)a.n_beads;
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
        const double *xc = a.x + rows[q] * n3;
        for (int k = threadIdx.x; k < n3; k += THREADS) row_lds[k * ROWS + q] = xc[k];
    }
    __syncthreads();
    PairResidRows<ROWS> f;
    f.xl = row_lds; f.I = a.I; f.J = a.J; f.ys = a.ys;
    double total[ROWS];
    block_sum_rows<ROWS, U, THREADS>(f, g.D, g.H, sc, round, total);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
            if (row0 + q >= g.C) break;
            if (g.memo_sum) *row_memo_slot(g, row0 + q) = total[q];
            out[row0 + q] = row_result(g, row0 + q, total[q]);
        }
    }
}

// ---- chi^2 with FEW chains: a workgroup per chain walks the whole pair list (0.3 ms at 1024
// beads, 1.2 ms at 2048, whatever the number of chains up to the number of CUs).  np.sum works
// in chunks of 8192 elements whose sums it adds one after the other, so the chunks are
// independent: here every (chunk, chain) is a workgroup, the chunk sums go to a workspace and a
// second launch adds them in order -- the same bits -- and applies the epilogue.  The
// coordinates are read through the caches (no LDS copy: any number of beads).
struct PairResidChunk {
    const double *xc;
    const int32_t *I;
    const int32_t *J;
    const double *ys;
    int64_t base;
    __device__ inline void operator()(int p, double (&v)[1]) const
    {
        const int64_t P = base + p;
        const int i = I[P], j = J[P];
        const double a = xc[3 * i] - xc[3 * j];
        const double b = xc[3 * i + 1] - xc[3 * j + 1];
        const double e = xc[3 * i + 2] - xc[3 * j + 2];
        const double s = (a * a + b * b) + e * e;
        const double d = sqrt_rn(s) - ys[P];
        v[0] = d * d;
    }
};

template <int U>
__global__ void __launch_bounds__(256)
pairdist_chi2_chunk_kernel(const PairArgs a, const RowGeom g, double *partial, int32_t nchunks)
{
    __shared__ BlockSumScratch sc;
    const int64_t c = blockIdx.y;
    const int k = blockIdx.x;
    if (g.skip && g.skip[c]) return;             // the memo holds this chain's sum (workgroup-uniform)
    int round = 0;
    PairResidChunk f;
    f.xc = a.x + c * 3 * a.n_beads; f.I = a.I; f.J = a.J; f.ys = a.ys;
    f.base = (int64_t)k * NPY_BUFSIZE;
    const int64_t left = (int64_t)g.D - f.base;
    const int len = left < NPY_BUFSIZE ? (int)left : NPY_BUFSIZE;
    double total[1];
    block_sum_rows<1, U, 256>(f, len, g.H, sc, round, total);    // 0.0 + this chunk's sum
    if (threadIdx.x == 0) partial[c * nchunks + k] = total[0];
}

// chunk sums in order, the memo, the epilogue; chi2_out (or null) receives the bare sums
__global__ void __launch_bounds__(256)
pairdist_chi2_join_kernel(const RowGeom g, const double *partial, int32_t nchunks, double *out, double *chi2_out)
{
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= g.C) return;
    double total;
    if (g.skip && g.skip[row]) {
        total = *row_memo_slot(g, row);
    } else {
        total = 0.0;                             // the reduction's identity
        const double *pr = partial + row * nchunks;
        int k = 0;
        for (; k + 8 <= nchunks; k += 8) {       // eight loads in flight, added in order
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = pr[k + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) total = total + v[u];
        }
        for (; k < nchunks; ++k) total = total + pr[k];
        if (g.memo_sum) *row_memo_slot(g, row) = total;
    }
    if (chi2_out) chi2_out[row] = total;
    if (out) out[row] = row_result(g, row, total);
}

// ---- the energy of HMCSampler.sample() (hmc.py:143,148,150) for the restraint posterior in
// ONE launch:  E = 0.5 np.sum(p**2) - log_prob,  log_prob = the Posterior's sum of its
// components in their order (binf/pdf/posteriors.py:147-151): the restraint likelihood
// (-0.5 chi^2 tau + N/2 log tau) and at most one isotropic Gaussian prior
// ((-0.5 k) np.sum((x - x0)**2)).  The per-step tier spends six launches on it (prior row
// sum, memo check, chi^2, term sum, kinetic row sum with the subtraction as its epilogue);
// a workgroup has the chain's coordinates in LDS anyway, so the two short row sums, the
// memo check and the few scalar operations ride along.  Every sum in numpy's order and
// every scalar operation as the per-step tier does it: the same bits.
struct PairEnergyArgs {
    const double *x;         // [C][3n]
    const double *p;         // [C][3n]
    const int32_t *I;
    const int32_t *J;
    const double *ys;
    double *memo_x;          // [2][C][3n] or null (no memo)
    uint8_t *memo_state;     // [2][C]
    double *energy;          // [C]
    double *log_prob;        // [C] or null
    double prior_scale;      // -0.5 k
    double prior_x0;
    int32_t has_prior;
    int32_t n_terms;         // the Posterior's components in its order (<= 4):
    int32_t term_kind[4];    //   0 the prior, 1 the likelihood, 2 / 3 a constant of the move
    const double *extra[2];  // constants per chain [C], or null: extra_scalar
    double extra_scalar[2];
    int32_t n_beads;
    int32_t H_d;             // tree height of a 3n-element sum
    const double *chi2_in;   // [C] or null: the restraints' chi^2 already summed (few chains: by chunks)
};

template <int ROWS>
struct ShiftSqRows {         // (x - x0)**2 from the interleaved LDS copy
    const double *xl;
    double x0;
    __device__ inline void operator()(int i, double (&v)[ROWS]) const
    {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const double d = xl[i * ROWS + r] - x0;
            v[r] = d * d;
        }
    }
};

template <int ROWS>
struct SqRows {              // p**2 from HBM
    const double *row[ROWS];
    __device__ inline void operator()(int i, double (&v)[ROWS]) const
    {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const double x = row[r][i];
            v[r] = x * x;
        }
    }
};

template <int ROWS, int U, int THREADS>
__global__ void __launch_bounds__(THREADS)
pairdist_energy_kernel(const PairEnergyArgs a, const RowGeom g)
{
    __shared__ BlockSumScratch sc;
    __shared__ int differs[ROWS][2];
    int round = 0;
    extern __shared__ double row_lds[];
    const int64_t row0 = (int64_t)blockIdx.x * ROWS;
    int64_t rows[ROWS];                          // the last workgroup repeats the last chain
#pragma unroll
    for (int q = 0; q < ROWS; ++q) rows[q] = (row0 + q < g.C) ? row0 + q : g.C - 1;
    const int n3 = 3 * a.n_beads;
    const bool given = a.chi2_in != nullptr;     // then the memo was served by the launches before this one
    const bool memo = a.memo_x != nullptr && !given;
    if (threadIdx.x < 2 * ROWS) differs[threadIdx.x >> 1][threadIdx.x & 1] = 0;
    __syncthreads();
    // coordinates to LDS; on the way, compared BIT FOR BIT with the memo's two entries
    // (row_memo_check_kernel, rowsum.hpp)
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
        const double *xc = a.x + rows[q] * n3;
        const double *m0 = memo ? a.memo_x + rows[q] * n3 : xc;
        const double *m1 = memo ? a.memo_x + (g.C + rows[q]) * n3 : xc;
        bool d0 = false, d1 = false;
        for (int k = threadIdx.x; k < n3; k += THREADS) {
            const double v = xc[k];
            row_lds[k * ROWS + q] = v;
            if (memo) {
                const long long bits = __double_as_longlong(v);
                d0 = d0 || bits != __double_as_longlong(m0[k]);
                d1 = d1 || bits != __double_as_longlong(m1[k]);
            }
        }
        if (d0) differs[q][0] = 1;
        if (d1) differs[q][1] = 1;
    }
    __syncthreads();
    bool hit[ROWS], all_hit = memo;
    int way[ROWS];
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
        const bool h0 = memo && !differs[q][0], h1 = memo && !differs[q][1];
        hit[q] = h0 || h1;
        way[q] = hit[q] ? (h0 ? 0 : 1) : (memo ? 1 - (a.memo_state[g.C + rows[q]] & 1) : 0);
        all_hit = all_hit && hit[q];
    }
    if (memo) {
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
            if (hit[q] || row0 + q >= g.C) continue;         // uniform
            double *m = a.memo_x + ((int64_t)way[q] * g.C + rows[q]) * n3;
            for (int k = threadIdx.x; k < n3; k += THREADS) m[k] = row_lds[k * ROWS + q];
        }
    }
    double chi2[ROWS];
    if (given) {
#pragma unroll
        for (int q = 0; q < ROWS; ++q) chi2[q] = a.chi2_in[rows[q]];
    } else if (!all_hit) {                       // uniform
        PairResidRows<ROWS> f;
        f.xl = row_lds; f.I = a.I; f.J = a.J; f.ys = a.ys;
        block_sum_rows<ROWS, U, THREADS>(f, g.D, g.H, sc, round, chi2);
    } else {
#pragma unroll
        for (int q = 0; q < ROWS; ++q) chi2[q] = g.memo_sum[(int64_t)way[q] * g.C + rows[q]];
    }
    double prior[ROWS], kin[ROWS];
#pragma unroll
    for (int q = 0; q < ROWS; ++q) prior[q] = 0.0;
    if (a.has_prior) {
        ShiftSqRows<ROWS> fp;
        fp.xl = row_lds; fp.x0 = a.prior_x0;
        block_sum_rows<ROWS, U, THREADS>(fp, n3, a.H_d, sc, round, prior);
    }
    SqRows<ROWS> fk;
#pragma unroll
    for (int q = 0; q < ROWS; ++q) fk.row[q] = a.p + rows[q] * n3;
    block_sum_rows<ROWS, U, THREADS>(fk, n3, a.H_d, sc, round, kin);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
            const int64_t row = row0 + q;
            if (row >= g.C) break;
            if (memo) {
                g.memo_sum[(int64_t)way[q] * g.C + row] = chi2[q];
                a.memo_state[row] = hit[q] ? 1 : 0;
                a.memo_state[g.C + row] = (uint8_t)way[q];
            }
            const double lp_lik = row_result(g, row, chi2[q]);
            const double lp_prior = a.prior_scale * prior[q];
            double lp = 0.0;                     // ((t0 + t1) + t2) + ..., binf_sum_terms_f64
            for (int k = 0; k < a.n_terms; ++k) {
                const int kind = a.term_kind[k];
                const double v = kind == 0 ? lp_prior
                               : kind == 1 ? lp_lik
                               : (a.extra[kind - 2] ? a.extra[kind - 2][row] : a.extra_scalar[kind - 2]);
                lp = (k == 0) ? v : lp + v;
            }
            if (a.log_prob) a.log_prob[row] = lp;
            a.energy[row] = 0.5 * kin[q] - lp;
        }
    }
}

// Restraint weight of one pair: w = (d - y) / d = 1 - y / d with d = |x_i - x_j|.
// IEEE sqrt + IEEE divide cost ~55 FP64 instructions per pair; instead
// 1/d = rsqrt(d^2) from the hardware seed (v_rsq_f64, ~24 good bits) polished
// by ONE Newton step (relative error ~1e-15; a second step measured 1.1e-15
// vs 6e-15 worst force error against numpy and 6 % more time), then one FMA.
// Used by BOTH force kernels, so the fused leapfrog and the per-step tier stay
// bit-identical to each other; against the numpy formulation the force is held
// to 1e-10.
__device__ inline double pair_weight(double d0, double d1, double d2, double y)
{
    const double s = (d0 * d0 + d1 * d1) + d2 * d2;
    double r = __builtin_amdgcn_rsq(s);
    r = r * __builtin_fma(-0.5 * s * r, r, 1.5);
    return __builtin_fma(-y, r, 1.0);
}

// Restraint force on bead i summed by FOUR adjacent lanes: lane quarter qt adds
// the pairs j in [qt*ceil(n/4), (qt+1)*ceil(n/4)) in ascending order, the four
// partial sums are joined as (f0 + f1) + (f2 + f3) by xor-shuffles 1 and 2 (all
// four lanes end up with the total).  With one lane per bead a 256-bead chain
// is one wave per SIMD and the loop is latency-bound; four lanes per bead give
// the scheduler four waves per SIMD.  sx: LDS copy of the chain's coordinates.
// Must be called by all lanes of the wave (shuffles).
__device__ inline void quartered_force(const double *sx, const double *ymat, int n, int i,
                                       bool valid, int qt, double x0, double x1, double x2,
                                       double &f0, double &f1, double &f2)
{
    const int per = (n + 3) >> 2;
    const int jb = qt * per;
    int je = jb + per;
    if (je > n) je = n;
    f0 = 0.0; f1 = 0.0; f2 = 0.0;
    if (valid) {
        for (int j = jb; j < je; ++j) {
            if (j == i) continue;
            const double d0 = x0 - sx[3 * j];
            const double d1 = x1 - sx[3 * j + 1];
            const double d2 = x2 - sx[3 * j + 2];
            const double w = pair_weight(d0, d1, d2, ymat[(int64_t)j * n + i]);
            f0 += w * d0;
            f1 += w * d1;
            f2 += w * d2;
        }
    }
    f0 = f0 + __shfl_xor(f0, 1, 64); f0 = f0 + __shfl_xor(f0, 2, 64);
    f1 = f1 + __shfl_xor(f1, 1, 64); f1 = f1 + __shfl_xor(f1, 2, 64);
    f2 = f2 + __shfl_xor(f2, 1, 64); f2 = f2 + __shfl_xor(f2, 2, 64);
}

// The same sum by ONE lane: four quarter accumulators advanced in lock-step
// (four independent chains of FMAs for the scheduler), joined in the same
// order -- bit-identical to quartered_force, so which variant a launch uses is
// a pure performance choice (many chains: one lane per bead; few: four).
__device__ inline void quartered_force_1lane(const double *sx, const double *ymat, int n, int i,
                                             double x0, double x1, double x2,
                                             double &f0, double &f1, double &f2)
{
    const int per = (n + 3) >> 2;
    double g[4][3];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) { g[qt][0] = 0.0; g[qt][1] = 0.0; g[qt][2] = 0.0; }
    for (int jj = 0; jj < per; ++jj) {
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            const int j = qt * per + jj;
            if (j < n && j != i) {
                const double d0 = x0 - sx[3 * j];
                const double d1 = x1 - sx[3 * j + 1];
                const double d2 = x2 - sx[3 * j + 2];
                const double w = pair_weight(d0, d1, d2, ymat[(int64_t)j * n + i]);
                g[qt][0] += w * d0;
                g[qt][1] += w * d1;
                g[qt][2] += w * d2;
            }
        }
    }
    f0 = (g[0][0] + g[1][0]) + (g[2][0] + g[3][0]);
    f1 = (g[0][1] + g[1][1]) + (g[2][1] + g[3][1]);
    f2 = (g[0][2] + g[1][2]) + (g[2][2] + g[3][2]);
}

// n_beads <= 1024, one lane per bead (256-thread workgroup per chain)
__global__ void __launch_bounds__(256)
pairdist_grad1_kernel(const double *x, const double *ymat, double tau,
                      const double *tau_chain, double *out, int32_t n_beads)
{
    extern __shared__ double sx[];
    const int n = n_beads;
    const int64_t c = blockIdx.x;
    const double *xc = x + c * 3 * (int64_t)n;
    const double t = tau_chain ? tau_chain[c] : tau;
    for (int k = threadIdx.x; k < 3 * n; k += 256) sx[k] = xc[k];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        double f0, f1, f2;
        quartered_force_1lane(sx, ymat, n, i, sx[3 * i], sx[3 * i + 1], sx[3 * i + 2], f0, f1, f2);
        double *o = out + c * 3 * (int64_t)n + 3 * i;
        o[0] = t * f0;
        o[1] = t * f1;
        o[2] = t * f2;
    }
}

// n_beads <= 1024: 1024-thread workgroup per chain, thread = (bead tile slot, quarter)
__global__ void __launch_bounds__(1024)
pairdist_grad4_kernel(const double *x, const double *ymat, double tau,
                      const double *tau_chain, double *out, int32_t n_beads)
{
    extern __shared__ double sx[];
    const int n = n_beads;
    const int64_t c = blockIdx.x;
    const double *xc = x + c * 3 * (int64_t)n;
    const double t = tau_chain ? tau_chain[c] : tau;
    for (int k = threadIdx.x; k < 3 * n; k += 1024) sx[k] = xc[k];
    __syncthreads();
    const int qt = threadIdx.x & 3;
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + (threadIdx.x >> 2);
        const bool iv = i < n;
        const double x0 = iv ? sx[3 * i] : 0.0, x1 = iv ? sx[3 * i + 1] : 0.0,
                     x2 = iv ? sx[3 * i + 2] : 0.0;
        double f0, f1, f2;
        quartered_force(sx, ymat, n, i, iv, qt, x0, x1, x2, f0, f1, f2);
        if (iv && qt == 0) {
            double *o = out + c * 3 * (int64_t)n + 3 * i;
            o[0] = t * f0;
            o[1] = t * f1;
            o[2] = t * f2;
        }
    }
}

// n_beads > 1024: one lane per bead, coordinates staged tile by tile.
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
                    const double w = pair_weight(a, b, e, ymat[(int64_t)j * n_beads + i]);
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

// ---------------------------------------------------------------------------
// Fused leapfrog for the restraint posterior: the whole _leapfrog() of
// binf/samplers/hmc.py:92-125 in one launch, one 1024-thread workgroup per chain.
// Four adjacent lanes share a bead (slot, slot + 256, ...; position and momentum
// in registers, identical in the four lanes); every bead position is mirrored in
// LDS for the all-pairs force loop.  The force is evaluated exactly like
// pairdist_grad4_kernel (quartered_force) and the
// gradient is assembled like the Posterior does (component terms in sorted-name
// order), so the result is bit-identical to the per-step generic tier.
// ---------------------------------------------------------------------------
struct PairLeapArgs {
    double *q;               // [C x 3n] in/out
    const double *q_from;    // [C x 3n] start positions when they are not to be read from q, or null
    double *p;               // [C x 3n] in/out
    const double *ymat;      // [n x n]
    const double *ypk;       // sym kernels: the targets in lane order (sym_pack_targets_kernel), or null
    const double *tau_chain; // [C] or null
    const double *dt_chain;  // [C] or null
    double tau;
    double timestep;
    double prior_k;
    double prior_x0;
    int32_t has_prior;
    int32_t prior_first;     // prior term added before the likelihood term
    int32_t nsteps;
    int32_t n_beads;
    int64_t n_chains;        // sym kernels: a workgroup walks chains blockIdx.x, + gridDim.x, ...
};

template <int NB, bool FMA, int LANES>
__global__ void __launch_bounds__(256 * LANES) pairdist_leapfrog_kernel(const PairLeapArgs a)
{
    extern __shared__ double sx[];               // [n_beads][3]
    const int n = a.n_beads;
    const int64_t c = blockIdx.x;
    double *qc = a.q + c * 3 * (int64_t)n;
    const double *qs = (a.q_from ? a.q_from : a.q) + c * 3 * (int64_t)n;
    double *pc = a.p + c * 3 * (int64_t)n;
    const double tau = a.tau_chain ? a.tau_chain[c] : a.tau;
    const double dt = a.dt_chain ? a.dt_chain[c] : a.timestep;
    const double hdt = 0.5 * dt;
    const int qt = (LANES == 4) ? (threadIdx.x & 3) : 0;              // quarter of the j-range
    const int slot = (LANES == 4) ? (threadIdx.x >> 2) : threadIdx.x;  // bead inside a tile of 256

    // the four lanes of a bead hold identical copies of its position / momentum.  Four
    // tiles of beads with four lanes each (769 .. 1024 beads, a 1024-thread workgroup:
    // 128 VGPRs) do not fit q AND p in registers -- 28 bytes per lane went to scratch --
    // so there the momentum lives in LDS behind the positions: every lane of a bead
    // writes the same value to the same slot and reads it back itself.
    constexpr bool PLDS = (NB == 4 && LANES == 4);
    double *sp = sx + 3 * n;                     // [n_beads][3], PLDS only
    double q[NB][3], p[PLDS ? 1 : NB][3];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = slot + 256 * b;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            q[b][ax] = (i < n) ? qs[3 * i + ax] : 0.0;
            const double pv = (i < n) ? pc[3 * i + ax] : 0.0;
            if (PLDS) { if (i < n) sp[3 * i + ax] = pv; }
            else p[b][ax] = pv;
        }
    }
    auto get_p = [&](int b, int ax) -> double {
        const int i = slot + 256 * b;
        return PLDS ? ((i < n) ? sp[3 * i + ax] : 0.0) : p[PLDS ? 0 : b][ax];
    };
    auto set_p = [&](int b, int ax, double v) {
        const int i = slot + 256 * b;
        if (PLDS) { if (i < n) sp[3 * i + ax] = v; }
        else p[PLDS ? 0 : b][ax] = v;
    };
    auto publish = [&]() {
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int i = slot + 256 * b;
            if (i < n && qt == 0) {
                sx[3 * i] = q[b][0];
                sx[3 * i + 1] = q[b][1];
                sx[3 * i + 2] = q[b][2];
            }
        }
        __syncthreads();
    };
    // p -= step * gradient(q)
    auto kick = [&](double step) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int i = slot + 256 * b;
            double f[3];
            if (LANES == 4) {
                quartered_force(sx, a.ymat, n, i, i < n, qt, q[b][0], q[b][1], q[b][2],
                                f[0], f[1], f[2]);
            } else {
                f[0] = f[1] = f[2] = 0.0;
                if (i < n)
                    quartered_force_1lane(sx, a.ymat, n, i, q[b][0], q[b][1], q[b][2],
                                          f[0], f[1], f[2]);
            }
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const double gl = tau * f[ax];
                double g = gl;
                if (a.has_prior) {
                    const double gp = a.prior_k * (q[b][ax] - a.prior_x0);
                    g = a.prior_first ? gp + gl : gl + gp;
                }
                const double pv = get_p(b, ax);
                set_p(b, ax, FMA ? __builtin_fma(-step, g, pv) : pv - step * g);
            }
        }
    };
    auto drift = [&]() {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax)
            {
                const double pv = get_p(b, ax);
                q[b][ax] = FMA ? __builtin_fma(pv, dt, q[b][ax]) : q[b][ax] + pv * dt;
            }
    };

    publish();
    kick(hdt);                                           // hmc.py:116
    for (int s = 0; s < a.nsteps - 1; ++s) {             // hmc.py:118-120
        drift();
        publish();
        kick(dt);
    }
    drift();                                             // hmc.py:122-123
    publish();
    kick(hdt);

#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = slot + 256 * b;
        if (i < n && qt == 0) {
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                qc[3 * i + ax] = q[b][ax];
                pc[3 * i + ax] = get_p(b, ax);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// 32 <= n_beads <= 256: every UNORDERED pair once, its target distance in a register.
//
// The all-pairs loops above give each lane the pairs (i, j = ...) of ITS bead and
// read y[j][i] from memory for each: one CU streams the whole [n x n] target
// matrix (512 KiB at n = 256) per force evaluation, and that stream -- not the
// arithmetic -- is what a workgroup waits for (scripts/pairforce_probe.hip: 30 us
// per evaluation with the loads, 10 us without, at any number of chains).  Here
// one workgroup owns a chain and computes each pair {i, j} ONCE:
//
//  * beads in NBLK = ceil(n / 64) blocks of 64; the unordered block pairs go to
//    NBLK^2 waves (n = 256: 16 waves, a 1024-thread workgroup): waves 0..NBLK-1 the
//    diagonal blocks (b, b), the others the off-diagonal pairs (bi < bj), two
//    waves each (the halves h = 0, 1 of the partner range);
//  * lane l of a wave is ROW bead i = 64 bi + l for all of its 32 steps; at step
//    k its partner is COLUMN bead j = 64 bj + (l + off + k) mod 64 (off = 32 h,
//    or 1 on the diagonal, where step 31 -- partner l + 32 -- is done by lanes
//    0-31 only).  At every step the 64 lanes hold 64 different partners;
//  * the 32 target distances of a lane never change: they are loaded ONCE per
//    launch (tile rows through LDS, coalesced) and stay in 64 VGPRs -- the fused
//    leapfrog reads the target matrix once per trajectory instead of L + 1 times;
//  * one pair weight serves both beads: w (x_i - x_j) is added to the lane's own
//    sum F (row bead) and subtracted from a second sum R that belongs to the
//    partner bead; since the partner advances by one column per step, R
//    is passed to the neighbouring lane after every step (wave_rol:1), so it
//    stays with its bead.  Half the arithmetic of the one-sided loops;
//  * a bead's 8 partial sums (partner block b' = 0..3: the diagonal wave's F and
//    R for b' = b, else the two half-waves' sums) are added in that order by the
//    bead's owner thread (threads 0-255) -- a fixed order that depends on n
//    only, so results are bit-identical for any number of chains, and the fused
//    leapfrog is bit-identical to the per-step tier (both use this scheme for
//    n <= 256; the one-sided kernels above serve larger n).
// ---------------------------------------------------------------------------
constexpr int SYM_MAX_BEADS = 256;
constexpr int SYM_MIN_BEADS = 32;        // fewer: most of the 32 steps would be masked (one-sided loops)
constexpr int SYM_STEPS = 32;
constexpr int SYM_ROWS = 8;              // rows of a wave's 64 x 64 target tile staged at a time

// NBLK = ceil(n / 64) blocks of 64 beads -> NBLK^2 waves (NBLK diagonal blocks one wave
// each, NBLK (NBLK - 1) / 2 off-diagonal pairs two waves each): a 64- / 256- / 576- /
// 1024-thread workgroup.  A wave computes the same partial sums whatever NBLK is, a
// bead's partials are added in slot order, and the slots of blocks that do not exist
// would hold +0.0 and come last -- so choosing NBLK from n (small workgroups for few
// beads: 16 / 4 chains per CU at a time instead of 1) does not change a bit.
template <int NBLK>
struct SymShared {
    double sx[NBLK][3][128];             // positions [block][axis][slot]; slots 64-127 repeat 0-63
                                         // (a block's three axes within one ds_read2 offset range)
    union {
        double part[2 * NBLK][3][64 * NBLK];     // partial forces [partner block * 2 + k][axis][bead]
        double ytile[NBLK * NBLK][SYM_ROWS][64]; // launch prologue only
    } u;
};

__device__ inline double wave_rol1(double v)     // lane l takes lane (l + 1) mod 64's value
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x134, 0xf, 0xf, false);   // old = src: no zeroing mov
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x134, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

struct SymRole {
    int bi, bj, h, off;
    bool diag;
};

template <int NBLK>
__device__ inline SymRole sym_role()
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    SymRole r;
    if (wave < NBLK) {
        r.bi = wave; r.bj = wave; r.h = 0; r.off = 1; r.diag = true;
    } else {
        int rem = (wave - NBLK) >> 1;              // pairs in the order (0,1) (0,2) .. (1,2) ..
        r.h = (wave - NBLK) & 1;
        r.bi = 0;
        while (rem >= NBLK - 1 - r.bi) { rem -= NBLK - 1 - r.bi; ++r.bi; }
        r.bj = r.bi + 1 + rem;
        r.off = 32 * r.h;
        r.diag = false;
    }
    return r;
}

// The lane's 32 target distances y[k] = ymat[64 bi + l][64 bj + (l + off + k) mod 64]
// (0 where a bead does not exist), and the bit mask of its pairs that do exist.
template <int NBLK, bool PK>
__device__ inline void sym_load_targets(double (&y)[SYM_STEPS], unsigned &live, SymShared<NBLK> &sh,
                                        const double *ymat, const double *ypk, int n,
                                        const SymRole &ro)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if constexpr (PK) {
        // packed by sym_pack_targets_kernel: step k of wave w, lane l at ypk[(32 w + k) 64 + l]
        // -- 32 independent coalesced loads, no exchange through LDS
        const double *src = ypk + (int64_t)wave * SYM_STEPS * 64 + lane;
#pragma unroll
        for (int k = 0; k < SYM_STEPS; ++k) y[k] = src[k * 64];
    } else {
    // The tile goes through a scratch area of the wave's own, SYM_ROWS rows at a
    // time (lane = column on the way in: coalesced 512-byte rows; lane = row on the
    // way out), the next rows' loads in flight meanwhile.  No other wave touches the
    // scratch: LDS serves a wave's instructions in order, so wave-level fences (no
    // s_barrier) are all the synchronisation the exchange needs.
    const int col = 64 * ro.bj + lane;
    double nxt[SYM_ROWS];
    auto fetch = [&](int r0) {
#pragma unroll
        for (int r = 0; r < SYM_ROWS; ++r) {
            const int row = 64 * ro.bi + r0 + r;
            nxt[r] = (row < n && col < n) ? ymat[(int64_t)row * n + col] : 0.0;
        }
    };
    fetch(0);
    for (int r0 = 0; r0 < 64; r0 += SYM_ROWS) {
#pragma unroll
        for (int r = 0; r < SYM_ROWS; ++r) sh.u.ytile[wave][r][lane] = nxt[r];
        if (r0 + SYM_ROWS < 64) fetch(r0 + SYM_ROWS);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane >= r0 && lane < r0 + SYM_ROWS) {
            const double *rowp = sh.u.ytile[wave][lane - r0];
#pragma unroll
            for (int k = 0; k < SYM_STEPS; ++k) y[k] = rowp[(lane + ro.off + k) & 63];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();                     // the scratch shares its LDS with the partial sums
    }
    live = 0;
    const int i = 64 * ro.bi + lane;
#pragma unroll
    for (int k = 0; k < SYM_STEPS; ++k) {
        const int j = 64 * ro.bj + ((lane + ro.off + k) & 63);
        const bool ok = i < n && j < n && !(ro.diag && k == SYM_STEPS - 1 && lane >= 32);
        live |= ok ? (1u << k) : 0u;
    }
}

// The targets in the order the waves of the sym kernels hold them: a function of
// (ymat, n) alone, written once per model and read by every launch after that
// (256 beads: 256 KiB instead of a 512 KiB matrix walked tile by tile through LDS).
template <int NBLK>
__global__ void __launch_bounds__(1024) sym_pack_targets_kernel(const double *ymat, double *ypk, int n)
{
    const SymRole ro = sym_role<NBLK>();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = 64 * ro.bi + lane;
    for (int k = 0; k < SYM_STEPS; ++k) {
        const int j = 64 * ro.bj + ((lane + ro.off + k) & 63);
        ypk[((int64_t)wave * SYM_STEPS + k) * 64 + lane] =
            (i < n && j < n) ? ymat[(int64_t)i * n + j] : 0.0;
    }
}

// One force evaluation: this wave's partial sums to sh.u.part.  Positions must be
// in sh.sx (both copies); the caller synchronises before reading the partials.
// FULL: n == 64 NBLK, every pair exists except the diagonal waves' last half step.
template <bool FULL, int NBLK>
__device__ inline void sym_partials(const double (&y)[SYM_STEPS], unsigned live, SymShared<NBLK> &sh,
                                    int n, const SymRole &ro)
{
    const int lane = threadIdx.x & 63;
    double F0 = 0.0, F1 = 0.0, F2 = 0.0, R0 = 0.0, R1 = 0.0, R2 = 0.0;
    if (64 * ro.bi < n && 64 * ro.bj < n) {            // wave-uniform
        const double x0 = sh.sx[ro.bi][0][lane], x1 = sh.sx[ro.bi][1][lane],
                     x2 = sh.sx[ro.bi][2][lane];
        const double *pj0 = &sh.sx[ro.bj][0][lane + ro.off];
        const double *pj1 = &sh.sx[ro.bj][1][lane + ro.off];
        const double *pj2 = &sh.sx[ro.bj][2][lane + ro.off];
#pragma unroll
        for (int k = 0; k < SYM_STEPS; ++k) {
            const double d0 = x0 - pj0[k], d1 = x1 - pj1[k], d2 = x2 - pj2[k];
            // pair_weight() with the squared distance and both bookings contracted to
            // FMAs (24 instead of 29 VALU instructions per pair; the force is held to
            // 1e-10 of numpy's, not to its bits, in either arithmetic mode)
            const double s2 = __builtin_fma(d2, d2, __builtin_fma(d1, d1, d0 * d0));
            double r = __builtin_amdgcn_rsq(s2);
            r = r * __builtin_fma(-0.5 * s2 * r, r, 1.5);
            double w = __builtin_fma(-y[k], r, 1.0);
            if (FULL) {
                if (k == SYM_STEPS - 1) w = (ro.diag && lane >= 32) ? 0.0 : w;
            } else {
                w = ((live >> k) & 1u) ? w : 0.0;      // also discards the NaN of a 0-distance ghost
            }
            F0 = __builtin_fma(w, d0, F0); F1 = __builtin_fma(w, d1, F1); F2 = __builtin_fma(w, d2, F2);
            R0 = wave_rol1(__builtin_fma(-w, d0, R0));  // the partner's share is -w d; then R moves on
            R1 = wave_rol1(__builtin_fma(-w, d1, R1));
            R2 = wave_rol1(__builtin_fma(-w, d2, R2));
        }
    }
    const int slotF = ro.diag ? 2 * ro.bi : 2 * ro.bj + ro.h;
    const int slotR = ro.diag ? 2 * ro.bi + 1 : 2 * ro.bi + ro.h;
    const int bF = 64 * ro.bi + lane;
    const int bR = 64 * ro.bj + ((lane + ro.off + SYM_STEPS) & 63);
    sh.u.part[slotF][0][bF] = F0; sh.u.part[slotF][1][bF] = F1; sh.u.part[slotF][2][bF] = F2;
    sh.u.part[slotR][0][bR] = R0; sh.u.part[slotR][1][bR] = R1; sh.u.part[slotR][2][bR] = R2;
}

// Owner thread t (< 64 NBLK) of bead t: the bead's force, partials added in slot order.
template <int NBLK>
__device__ inline void sym_reduce(const SymShared<NBLK> &sh, int t, double (&f)[3])
{
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        double v = sh.u.part[0][ax][t];
#pragma unroll
        for (int sl = 1; sl < 2 * NBLK; ++sl) v = v + sh.u.part[sl][ax][t];
        f[ax] = v;
    }
}

template <int NBLK>
__device__ inline void sym_publish(SymShared<NBLK> &sh, int t, const double (&q)[3])
{
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        sh.sx[t >> 6][ax][t & 63] = q[ax];
        sh.sx[t >> 6][ax][(t & 63) + 64] = q[ax];
    }
}

// A workgroup loads its target distances once and then walks chains blockIdx.x,
// blockIdx.x + gridDim.x, ...: the launchers start as many workgroups as the chip
// holds at a time (126 VGPRs: 16 waves per CU), so with more chains than that the
// target load is paid once per workgroup, not once per chain.
// __launch_bounds__(1024) whatever NBLK: it is what keeps the kernel within 128 VGPRs.
template <bool FULL, int NBLK, bool PK>
__global__ void __launch_bounds__(1024)
pairdist_grad_sym_kernel(const double *x, const double *ymat, const double *ypk, double tau,
                         const double *tau_chain, double *out, int32_t n_beads, int64_t n_chains)
{
    __shared__ SymShared<NBLK> sh;
    const int n = n_beads, t = threadIdx.x;
    const SymRole ro = sym_role<NBLK>();
    double y[SYM_STEPS];
    unsigned live;
    sym_load_targets<NBLK, PK>(y, live, sh, ymat, ypk, n, ro);
    for (int64_t c = blockIdx.x; c < n_chains; c += gridDim.x) {
        const double *xc = x + c * 3 * (int64_t)n;
        if (t < 64 * NBLK) {
            double q[3];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) q[ax] = (t < n) ? xc[3 * t + ax] : 0.0;
            sym_publish<NBLK>(sh, t, q);
        }
        __syncthreads();
        sym_partials<FULL, NBLK>(y, live, sh, n, ro);
        __syncthreads();
        if (t < n) {
            double f[3];
            sym_reduce<NBLK>(sh, t, f);
            const double tc = tau_chain ? tau_chain[c] : tau;
            double *o = out + c * 3 * (int64_t)n + 3 * t;
            o[0] = tc * f[0]; o[1] = tc * f[1]; o[2] = tc * f[2];
        }
    }
}

// The whole _leapfrog() (binf/samplers/hmc.py:92-125) in one launch with the
// scheme above: owner threads keep q and p of their bead in registers.
template <bool FMA, bool FULL, int NBLK, bool PK>
__global__ void __launch_bounds__(1024) pairdist_leapfrog_sym_kernel(const PairLeapArgs a)
{
    __shared__ SymShared<NBLK> sh;
    const int n = a.n_beads, t = threadIdx.x;
    const SymRole ro = sym_role<NBLK>();
    const bool owner = t < n;
    double y[SYM_STEPS];
    unsigned live;
    sym_load_targets<NBLK, PK>(y, live, sh, a.ymat, a.ypk, n, ro);
    for (int64_t c = blockIdx.x; c < a.n_chains; c += gridDim.x) {
        double *qc = a.q + c * 3 * (int64_t)n;
        const double *qs = (a.q_from ? a.q_from : a.q) + c * 3 * (int64_t)n;
        double *pc = a.p + c * 3 * (int64_t)n;
        const double tau = a.tau_chain ? a.tau_chain[c] : a.tau;
        const double dt = a.dt_chain ? a.dt_chain[c] : a.timestep;
        const double hdt = 0.5 * dt;
        double q[3] = {0.0, 0.0, 0.0}, p[3] = {0.0, 0.0, 0.0};
        if (owner) {
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) { q[ax] = qs[3 * t + ax]; p[ax] = pc[3 * t + ax]; }
        }
        if (t < 64 * NBLK) sym_publish<NBLK>(sh, t, q);
        __syncthreads();
        // nsteps + 1 force evaluations: half kick, (nsteps - 1) x [drift, kick], drift, half kick
        for (int e = 0; e <= a.nsteps; ++e) {
            sym_partials<FULL, NBLK>(y, live, sh, n, ro);
            __syncthreads();
            if (owner) {
                double f[3];
                sym_reduce<NBLK>(sh, t, f);
                const double step = (e == 0 || e == a.nsteps) ? hdt : dt;
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const double gl = tau * f[ax];
                    double g = gl;
                    if (a.has_prior) {
                        const double gp = a.prior_k * (q[ax] - a.prior_x0);
                        g = a.prior_first ? gp + gl : gl + gp;
                    }
                    p[ax] = FMA ? __builtin_fma(-step, g, p[ax]) : p[ax] - step * g;
                    if (e < a.nsteps)
                        q[ax] = FMA ? __builtin_fma(p[ax], dt, q[ax]) : q[ax] + p[ax] * dt;
                }
                if (e < a.nsteps) sym_publish<NBLK>(sh, t, q);
            }
            __syncthreads();
        }
        if (owner) {
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) { qc[3 * t + ax] = q[ax]; pc[3 * t + ax] = p[ax]; }
        }
    }
}

// ---------------------------------------------------------------------------
// 256 < n_beads <= 1024: every UNORDERED pair once as well -- the "ring" scheme.
//
// The scheme above keeps a bead's 2 NBLK partial sums in LDS and a wave's 32 target
// distances in registers for the whole launch; neither scales past 256 beads (512
// beads: 196 KB of partial sums, 128 targets per lane).  Here a workgroup of NBLK =
// ceil(n / 64) waves owns a chain, wave b = the 64 ROW beads of block b, and the block
// pairs are walked in PHASES that every wave takes together:
//
//   phase 0       the diagonal tile (b, b): 31.5 steps, partner l + 1 + m (as above)
//   phase s >= 1  the tile (b, (b + s) mod NBLK): 64 steps, partner column l + m;
//                 s runs to NBLK / 2.  For an even NBLK the last phase pairs antipodal
//                 blocks, which see each other from both sides: the lower block takes
//                 the column offsets 0..31, the upper one 1..32 (32 steps each) -- an
//                 exact cover of the tile.
//
//  * In every phase each COLUMN block is visited by exactly one wave.  A lane adds
//    w (x_i - x_j) to its row bead's sum F (a register for the whole evaluation) and
//    subtracts it from the partner's sum R, which moves on by one lane per step
//    (wave_rol1) and so stays with its bead; at the end of a phase R goes to ONE LDS
//    slot per bead, and after a barrier the bead's owner adds it to a second register
//    G.  The force is F + G: F summed in step order, G in phase order -- a fixed order
//    that depends on n only, so results are bit-identical for any number of chains and
//    the fused leapfrog is bit-identical to the per-step tier.  Half the arithmetic of
//    the one-sided loops, one barrier per phase (5 at 512 beads, 9 at 1024).
//  * The targets cannot stay in registers: they are read again for every force
//    evaluation, 8 steps ahead of their use, from the packed array (ring_pack_targets_
//    kernel: step t of wave b, lane l at ypk[(b T + t) 64 + l] -- coalesced, 1 MiB at
//    512 beads, shared by all chains and resident in L2).
// ---------------------------------------------------------------------------
constexpr int RING_MAX_BEADS = 1024;     // a workgroup (<= 16 waves) per chain
constexpr int TILES_MAX_BEADS = 8192;    // a wave per tile: any number of blocks (packed targets: 256 MiB here)
constexpr int RING_CHUNK = 8;            // steps whose targets are in flight / in use at a time
                                         // (4: -2 %, 16: -8 % at 512 .. 1024 beads, same-box A/B)

__host__ __device__ inline int ring_steps(int nblk)      // steps a wave walks per force evaluation
{
    return (nblk & 1) ? 32 + 64 * (nblk / 2) : 64 * (nblk / 2);
}

struct RingPhase {
    int bj, off, steps;
};

__host__ __device__ inline RingPhase ring_phase(int nblk, int bi, int s)
{
    RingPhase ph;
    if (s == 0) {
        ph.bj = bi; ph.off = 1; ph.steps = 32;
    } else {
        const bool half = !(nblk & 1) && s == nblk / 2;
        ph.bj = bi + s; if (ph.bj >= nblk) ph.bj -= nblk;
        ph.off = (half && bi >= nblk / 2) ? 1 : 0;
        ph.steps = half ? 32 : 64;
    }
    return ph;
}

template <int NBT>
struct RingShared {
    double sx[NBT][3][128];              // positions [block][axis][slot]; slots 64-127 repeat 0-63
    double racc[2][NBT][3][64];          // a phase's column-side sums [buffer][block][axis][bead]
};

// The targets in the order the waves of the ring kernels read them.
__global__ void __launch_bounds__(256) ring_pack_targets_kernel(const double *ymat, double *ypk, int n, int nblk)
{
    const int bi = blockIdx.x, lane = threadIdx.x & 63;
    const int total = ring_steps(nblk);
    const int i = 64 * bi + lane;
    for (int t = threadIdx.x >> 6; t < total; t += blockDim.x >> 6) {
        const int s = t < 32 ? 0 : 1 + (t - 32) / 64;
        const int m = t < 32 ? t : (t - 32) % 64;
        const RingPhase ph = ring_phase(nblk, bi, s);
        const int j = 64 * ph.bj + ((lane + ph.off + m) & 63);
        ypk[((int64_t)bi * total + t) * 64 + lane] = (i < n && j < n) ? ymat[(int64_t)i * n + j] : 0.0;
    }
}

// RING_CHUNK steps of a phase.  MASK: some of this tile's pairs do not exist (the last half
// step of a diagonal tile; beads beyond n when n is not a multiple of 64 and the row or the
// column block is the last one) -- bit k of `live` says whether the pair of step m0 + k does.
template <bool MASK>
__device__ inline void ring_chunk(const double (&y)[RING_CHUNK], const double *pj0, const double *pj1,
                                  const double *pj2, int m0, double x0, double x1, double x2,
                                  unsigned live,
                                  double &F0, double &F1, double &F2, double &R0, double &R1, double &R2)
{
#pragma unroll
    for (int k = 0; k < RING_CHUNK; ++k) {
        const int m = m0 + k;
        const double d0 = x0 - pj0[m], d1 = x1 - pj1[m], d2 = x2 - pj2[m];
        const double s2 = __builtin_fma(d2, d2, __builtin_fma(d1, d1, d0 * d0));
        double r = __builtin_amdgcn_rsq(s2);
        r = r * __builtin_fma(-0.5 * s2 * r, r, 1.5);
        double w = __builtin_fma(-y[k], r, 1.0);
        if (MASK) w = ((live >> k) & 1u) ? w : 0.0;     // also discards the NaN of a 0-distance ghost pair
        F0 = __builtin_fma(w, d0, F0); F1 = __builtin_fma(w, d1, F1); F2 = __builtin_fma(w, d2, F2);
        R0 = wave_rol1(__builtin_fma(-w, d0, R0));
        R1 = wave_rol1(__builtin_fma(-w, d1, R1));
        R2 = wave_rol1(__builtin_fma(-w, d2, R2));
    }
}

// One force evaluation of the chain whose positions are in sh.sx (both copies): the force on
// this thread's bead (wave = block, lane = bead in the block) in f.  Every wave of the
// workgroup must call it; it ends behind the last phase's barrier, so the caller may
// overwrite sh.sx at once (and must synchronise before the next evaluation).
template <bool FULL, int NBT>
__device__ inline void ring_force(RingShared<NBT> &sh, const double *ypk, int n, int nblk, int &buf,
                                  double (&f)[3])
{
    const int bi = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int total = ring_steps(nblk);
    const double *yp = ypk + (int64_t)bi * total * 64 + lane;
    const double x0 = sh.sx[bi][0][lane], x1 = sh.sx[bi][1][lane], x2 = sh.sx[bi][2][lane];
    const bool row_ok = 64 * bi + lane < n;
    double F0 = 0.0, F1 = 0.0, F2 = 0.0, G0 = 0.0, G1 = 0.0, G2 = 0.0;
    double y[RING_CHUNK], yn[RING_CHUNK];
#pragma unroll
    for (int k = 0; k < RING_CHUNK; ++k) y[k] = yp[k * 64];
    int t = 0;
    const int nph = nblk / 2;
    for (int s = 0; s <= nph; ++s) {
        const RingPhase ph = ring_phase(nblk, bi, s);
        const double *pj0 = &sh.sx[ph.bj][0][lane + ph.off];
        const double *pj1 = &sh.sx[ph.bj][1][lane + ph.off];
        const double *pj2 = &sh.sx[ph.bj][2][lane + ph.off];
        // which of the lane's pairs exist in this tile: bit m <-> step m (wave-uniform test)
        const bool mask = s == 0 || (!FULL && (bi == nblk - 1 || ph.bj == nblk - 1));
        unsigned long long live = ~0ull;
        if (mask) {
            const int col_lim = n - 64 * ph.bj;                               // columns that exist
            const unsigned long long cols = col_lim >= 64 ? ~0ull : ((1ull << col_lim) - 1ull);
            const int rot = (lane + ph.off) & 63;                             // bit m <- column (rot + m) mod 64
            live = rot ? ((cols >> rot) | (cols << (64 - rot))) : cols;
            if (!row_ok) live = 0ull;
            if (s == 0 && lane >= 32) live &= ~(1ull << 31);                  // partner l + 32: lanes 0-31 only
        }
        double P0 = 0.0, P1 = 0.0, P2 = 0.0, R0 = 0.0, R1 = 0.0, R2 = 0.0;   // this TILE's row / column sums
        for (int m0 = 0; m0 < ph.steps; m0 += RING_CHUNK) {
            const bool more = t + RING_CHUNK < total;                         // wave-uniform
            if (more) {
#pragma unroll
                for (int k = 0; k < RING_CHUNK; ++k) yn[k] = yp[(int64_t)(t + RING_CHUNK + k) * 64];
            }
            if (mask) ring_chunk<true>(y, pj0, pj1, pj2, m0, x0, x1, x2, (unsigned)(live >> m0), P0, P1, P2, R0, R1, R2);
            else      ring_chunk<false>(y, pj0, pj1, pj2, m0, x0, x1, x2, 0u, P0, P1, P2, R0, R1, R2);
            if (more) {
#pragma unroll
                for (int k = 0; k < RING_CHUNK; ++k) y[k] = yn[k];
            }
            t += RING_CHUNK;
        }
        // a bead's force is the sum of its tiles' partial sums in phase order (row side F, column
        // side G) -- the same numbers whether one workgroup walks the phases (here) or every tile
        // is a wave of its own (pairdist_tiles_kernel below)
        F0 = s == 0 ? P0 : F0 + P0; F1 = s == 0 ? P1 : F1 + P1; F2 = s == 0 ? P2 : F2 + P2;
        // the column-side sums of this phase to their beads' slots; one wave per column block
        const int col = (lane + ph.off + ph.steps) & 63;
        sh.racc[buf][ph.bj][0][col] = R0; sh.racc[buf][ph.bj][1][col] = R1; sh.racc[buf][ph.bj][2][col] = R2;
        __syncthreads();
        G0 = G0 + sh.racc[buf][bi][0][lane]; G1 = G1 + sh.racc[buf][bi][1][lane]; G2 = G2 + sh.racc[buf][bi][2][lane];
        buf ^= 1;
    }
    f[0] = F0 + G0; f[1] = F1 + G1; f[2] = F2 + G2;
}

template <int NBT>
__device__ inline void ring_publish(RingShared<NBT> &sh, int t, const double (&q)[3])
{
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        sh.sx[t >> 6][ax][t & 63] = q[ax];
        sh.sx[t >> 6][ax][(t & 63) + 64] = q[ax];
    }
}

// One force evaluation per chain (the per-step tier's gradient): 64 NBLK threads per
// workgroup, a workgroup walks chains blockIdx.x, blockIdx.x + gridDim.x, ...
template <bool FULL, int NBT>
__global__ void __launch_bounds__(1024)
pairdist_grad_ring_kernel(const double *x, const double *ypk, double tau, const double *tau_chain,
                          double *out, int32_t n_beads, int32_t nblk, int64_t n_chains)
{
    __shared__ RingShared<NBT> sh;
    const int n = n_beads, t = threadIdx.x;
    int buf = 0;
    for (int64_t c = blockIdx.x; c < n_chains; c += gridDim.x) {
        const double *xc = x + c * 3 * (int64_t)n;
        double q[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) q[ax] = (t < n) ? xc[3 * t + ax] : 0.0;
        ring_publish<NBT>(sh, t, q);
        __syncthreads();
        double f[3];
        ring_force<FULL, NBT>(sh, ypk, n, nblk, buf, f);
        if (t < n) {
            const double tc = tau_chain ? tau_chain[c] : tau;
            double *o = out + c * 3 * (int64_t)n + 3 * t;
            o[0] = tc * f[0]; o[1] = tc * f[1]; o[2] = tc * f[2];
        }
    }
}

// The whole _leapfrog() (binf/samplers/hmc.py:92-125) in one launch with the ring scheme:
// every thread keeps q and p of its bead in registers.
template <bool FMA, bool FULL, int NBT>
__global__ void __launch_bounds__(1024) pairdist_leapfrog_ring_kernel(const PairLeapArgs a, int32_t nblk)
{
    __shared__ RingShared<NBT> sh;
    const int n = a.n_beads, t = threadIdx.x;
    const bool owner = t < n;
    int buf = 0;
    for (int64_t c = blockIdx.x; c < a.n_chains; c += gridDim.x) {
        double *qc = a.q + c * 3 * (int64_t)n;
        const double *qs = (a.q_from ? a.q_from : a.q) + c * 3 * (int64_t)n;
        double *pc = a.p + c * 3 * (int64_t)n;
        const double tau = a.tau_chain ? a.tau_chain[c] : a.tau;
        const double dt = a.dt_chain ? a.dt_chain[c] : a.timestep;
        const double hdt = 0.5 * dt;
        double q[3] = {0.0, 0.0, 0.0}, p[3] = {0.0, 0.0, 0.0};
        if (owner) {
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) { q[ax] = qs[3 * t + ax]; p[ax] = pc[3 * t + ax]; }
        }
        ring_publish<NBT>(sh, t, q);
        __syncthreads();
        // nsteps + 1 force evaluations: half kick, (nsteps - 1) x [drift, kick], drift, half kick
        for (int e = 0; e <= a.nsteps; ++e) {
            double f[3];
            ring_force<FULL, NBT>(sh, a.ypk, n, nblk, buf, f);
            if (owner) {
                const double step = (e == 0 || e == a.nsteps) ? hdt : dt;
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const double gl = tau * f[ax];
                    double g = gl;
                    if (a.has_prior) {
                        const double gp = a.prior_k * (q[ax] - a.prior_x0);
                        g = a.prior_first ? gp + gl : gl + gp;
                    }
                    p[ax] = FMA ? __builtin_fma(-step, g, p[ax]) : p[ax] - step * g;
                    if (e < a.nsteps)
                        q[ax] = FMA ? __builtin_fma(p[ax], dt, q[ax]) : q[ax] + p[ax] * dt;
                }
                if (e < a.nsteps) ring_publish<NBT>(sh, t, q);
            }
            __syncthreads();
        }
        if (owner) {
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) { qc[3 * t + ax] = q[ax]; pc[3 * t + ax] = p[ax]; }
        }
    }
}

// ---------------------------------------------------------------------------
// The same tiles with FEW chains: a workgroup per chain (above) leaves the chip idle
// when there are fewer chains than CUs -- 32 replicas of a 1000-bead model use an eighth
// of it.  Here every tile (row block b, phase s) is a wave of its own, whatever chain it
// belongs to: the tile's row-side and column-side sums go to a workspace
// (part[chain][tile][side][axis][bead in block]) and a second launch adds a bead's
// partial sums in the ring kernels' order -- the same bits -- and scales, kicks or drifts.
// ---------------------------------------------------------------------------
constexpr int TILE_WAVES = 4;            // tiles per workgroup

__host__ __device__ inline int ring_tiles(int nblk) { return nblk * (nblk / 2 + 1); }

template <bool FULL>
__global__ void __launch_bounds__(64 * TILE_WAVES)
pairdist_tiles_kernel(const double *x, const double *ypk, double *part, int32_t n, int32_t nblk)
{
    __shared__ double sxw[TILE_WAVES][3][128];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int nph = nblk / 2, ntiles = ring_tiles(nblk);
    const int tile = blockIdx.x * TILE_WAVES + wave;
    if (tile >= ntiles) return;                      // no workgroup barrier below
    const int bi = tile / (nph + 1), s = tile - bi * (nph + 1);
    const int64_t c = blockIdx.y;
    const double *xc = x + c * 3 * (int64_t)n;
    const RingPhase ph = ring_phase(nblk, bi, s);
    const int i = 64 * bi + lane, jb = 64 * ph.bj + lane;
    const bool row_ok = i < n;
    double x0 = 0.0, x1 = 0.0, x2 = 0.0;
    if (row_ok) { x0 = xc[3 * i]; x1 = xc[3 * i + 1]; x2 = xc[3 * i + 2]; }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const double v = jb < n ? xc[3 * jb + ax] : 0.0;
        sxw[wave][ax][lane] = v;
        sxw[wave][ax][lane + 64] = v;
    }
    // the wave's own scratch: LDS serves a wave's instructions in order, wave-level fences do
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double *pj0 = &sxw[wave][0][lane + ph.off];
    const double *pj1 = &sxw[wave][1][lane + ph.off];
    const double *pj2 = &sxw[wave][2][lane + ph.off];
    const bool mask = s == 0 || (!FULL && (bi == nblk - 1 || ph.bj == nblk - 1));
    unsigned long long live = ~0ull;
    if (mask) {
        const int col_lim = n - 64 * ph.bj;
        const unsigned long long cols = col_lim >= 64 ? ~0ull : ((1ull << col_lim) - 1ull);
        const int rot = (lane + ph.off) & 63;
        live = rot ? ((cols >> rot) | (cols << (64 - rot))) : cols;
        if (!row_ok) live = 0ull;
        if (s == 0 && lane >= 32) live &= ~(1ull << 31);
    }
    const int t0 = s == 0 ? 0 : 32 + 64 * (s - 1);
    const double *yp = ypk + ((int64_t)bi * ring_steps(nblk) + t0) * 64 + lane;
    double y[RING_CHUNK], yn[RING_CHUNK];
#pragma unroll
    for (int k = 0; k < RING_CHUNK; ++k) y[k] = yp[k * 64];
    double P0 = 0.0, P1 = 0.0, P2 = 0.0, R0 = 0.0, R1 = 0.0, R2 = 0.0;
    for (int m0 = 0; m0 < ph.steps; m0 += RING_CHUNK) {
        const bool more = m0 + RING_CHUNK < ph.steps;
        if (more) {
#pragma unroll
            for (int k = 0; k < RING_CHUNK; ++k) yn[k] = yp[(int64_t)(m0 + RING_CHUNK + k) * 64];
        }
        if (mask) ring_chunk<true>(y, pj0, pj1, pj2, m0, x0, x1, x2, (unsigned)(live >> m0), P0, P1, P2, R0, R1, R2);
        else      ring_chunk<false>(y, pj0, pj1, pj2, m0, x0, x1, x2, 0u, P0, P1, P2, R0, R1, R2);
        if (more) {
#pragma unroll
            for (int k = 0; k < RING_CHUNK; ++k) y[k] = yn[k];
        }
    }
    double *o = part + ((c * ntiles + tile) * 2) * 3 * 64;
    const int col = (lane + ph.off + ph.steps) & 63;
    o[0 * 64 + lane] = P0; o[1 * 64 + lane] = P1; o[2 * 64 + lane] = P2;
    o[3 * 64 + col] = R0; o[4 * 64 + col] = R1; o[5 * 64 + col] = R2;
}

// A bead's force from the tiles' partial sums, in the ring kernels' order.  The loads of four
// tiles (three axes each) are issued before their sums are added one after the other: a loop
// that waits for every load cost 59 us per force evaluation at 8 chains of 4096 beads.
__device__ inline void tiles_force(const double *part, int64_t c, int bead, int nblk, double (&f)[3])
{
    constexpr int UN = 4;
    const int nph = nblk / 2, ntiles = ring_tiles(nblk);
    const int b = bead >> 6, l = bead & 63;
    const double *pc = part + c * ntiles * 2 * 3 * 64 + l;
    double F[3] = {0.0, 0.0, 0.0}, G[3] = {0.0, 0.0, 0.0};
    // row side: the tiles (b, s), s = 0 .. nph
    for (int s0 = 0; s0 <= nph; s0 += UN) {
        double v[UN][3];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int s = (s0 + u <= nph) ? s0 + u : nph;            // clamped: loaded, not added
            const double *pt = pc + ((int64_t)(b * (nph + 1) + s) * 2) * 192;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) v[u][ax] = pt[ax * 64];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (s0 + u > nph) break;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) F[ax] = (s0 + u == 0) ? v[u][ax] : F[ax] + v[u][ax];
        }
    }
    // column side: at phase s the block's partner is row block (b - s) mod NBLK
    for (int s0 = 0; s0 <= nph; s0 += UN) {
        double v[UN][3];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int s = (s0 + u <= nph) ? s0 + u : nph;
            int bi = b - s; if (bi < 0) bi += nblk;
            const double *pt = pc + ((int64_t)(bi * (nph + 1) + s) * 2 + 1) * 192;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) v[u][ax] = pt[ax * 64];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (s0 + u > nph) break;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) G[ax] = G[ax] + v[u][ax];
        }
    }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) f[ax] = F[ax] + G[ax];
}

__global__ void __launch_bounds__(256)
pairdist_tiles_grad_kernel(const double *part, double tau, const double *tau_chain, double *out,
                           int32_t n, int32_t nblk)
{
    const int bead = blockIdx.x * 256 + threadIdx.x;
    const int64_t c = blockIdx.y;
    if (bead >= n) return;
    double f[3];
    tiles_force(part, c, bead, nblk, f);
    const double tc = tau_chain ? tau_chain[c] : tau;
    double *o = out + c * 3 * (int64_t)n + 3 * bead;
    o[0] = tc * f[0]; o[1] = tc * f[1]; o[2] = tc * f[2];
}

// Force evaluation e of a trajectory: the kick (and, before the last one, the drift) of
// pairdist_leapfrog_ring_kernel's owner threads, on q and p in memory.
template <bool FMA>
__global__ void __launch_bounds__(256)
pairdist_tiles_update_kernel(const double *part, const PairLeapArgs a, const double *q_in, int32_t nblk, int32_t e)
{
    const int n = a.n_beads;
    const int bead = blockIdx.x * 256 + threadIdx.x;
    const int64_t c = blockIdx.y;
    if (bead >= n) return;
    double f[3];
    tiles_force(part, c, bead, nblk, f);
    const double tau = a.tau_chain ? a.tau_chain[c] : a.tau;
    const double dt = a.dt_chain ? a.dt_chain[c] : a.timestep;
    const double step = (e == 0 || e == a.nsteps) ? 0.5 * dt : dt;
    const int64_t o = c * 3 * (int64_t)n + 3 * bead;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        double q = q_in[o + ax], p = a.p[o + ax];
        const double gl = tau * f[ax];
        double g = gl;
        if (a.has_prior) {
            const double gp = a.prior_k * (q - a.prior_x0);
            g = a.prior_first ? gp + gl : gl + gp;
        }
        p = FMA ? __builtin_fma(-step, g, p) : p - step * g;
        if (e < a.nsteps) q = FMA ? __builtin_fma(p, dt, q) : q + p * dt;
        a.p[o + ax] = p;
        a.q[o + ax] = q;
    }
}

}  // namespace binf

using namespace binf;

// Few chains (less than ~4 workgroups of 256 threads per CU): four lanes per
// bead; many chains: one.  Both sum in the same order (bit-identical results).
static int lanes_per_bead(int64_t C) { return C < 1024 ? 4 : 1; }

// workgroups of the n <= 256 kernels: one per CU (each walks its share of the chains)
// development aid: BINF_PD_SYM=0 sends n <= 256 to the one-sided kernels as well
static bool sym_enabled()
{
    static std::atomic<int> on(-1);
    int v = on.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("BINF_PD_SYM");
        v = (e && e[0] == '0') ? 0 : 1;
        on.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}

static int cu_count()
{
    // CU count of the CURRENT device (the wrappers make the stream's device current),
    // cached per device id; atomics make the first calls of several threads harmless
    static std::atomic<int> cu_cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int cus = cu_cache[dev].load(std::memory_order_relaxed);
    if (cus <= 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        cu_cache[dev].store(cus, std::memory_order_relaxed);
    }
    return cus;
}

// workgroups of the n <= 256 kernels: as many as the chip holds at a time (16 waves per
// CU at 126 VGPRs: 16 / 4 / 1 / 1 workgroups of 1 / 4 / 9 / 16 waves); each walks its
// share of the chains
static unsigned sym_grid(int64_t C, int nblk)
{
    const int64_t slots = (int64_t)cu_count() * (nblk == 1 ? 16 : (nblk == 2 ? 4 : 1));
    return (unsigned)(C < slots ? C : slots);
}

template <int NBLK>
static void launch_grad_sym(const double *x, const double *ymat, const double *ypk, double precision,
                            const double *precision_chain, double *out, int64_t C, int64_t n,
                            hipStream_t st)
{
    const dim3 grid(sym_grid(C, NBLK)), block(64 * NBLK * NBLK);
    const bool full = n == 64 * NBLK;
#define GRAD_SYM(FULLV, PKV) \
    pairdist_grad_sym_kernel<FULLV, NBLK, PKV><<<grid, block, 0, st>>>(x, ymat, ypk, precision, \
                                                                       precision_chain, out, (int32_t)n, C)
    if (ypk) { if (full) GRAD_SYM(true, true); else GRAD_SYM(false, true); }
    else     { if (full) GRAD_SYM(true, false); else GRAD_SYM(false, false); }
#undef GRAD_SYM
}

template <int NBLK>
static void launch_leapfrog_sym(const PairLeapArgs &a, bool fma, hipStream_t st)
{
    const dim3 grid(sym_grid(a.n_chains, NBLK)), block(64 * NBLK * NBLK);
    const bool full = a.n_beads == 64 * NBLK;
#define LEAP_SYM(FMAV, FULLV) \
    do { \
        if (a.ypk) pairdist_leapfrog_sym_kernel<FMAV, FULLV, NBLK, true><<<grid, block, 0, st>>>(a); \
        else       pairdist_leapfrog_sym_kernel<FMAV, FULLV, NBLK, false><<<grid, block, 0, st>>>(a); \
    } while (0)
    if (fma) { if (full) LEAP_SYM(true, true); else LEAP_SYM(true, false); }
    else     { if (full) LEAP_SYM(false, true); else LEAP_SYM(false, false); }
#undef LEAP_SYM
}

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

static int32_t pairdist_logp_run(const double *x, const int32_t *pair_i, const int32_t *pair_j,
                                 const double *ys, double precision, const double *precision_chain,
                                 double *out, const uint8_t *skip, double *memo_chi2, int64_t C,
                                 int64_t n_beads, int64_t n_pairs, void *workspace,
                                 int64_t workspace_bytes, void *stream);

// chi^2 by chunks (pairdist_chi2_chunk_kernel) pays while a workgroup per chain leaves CUs idle and
// the pair list is long enough to split (development aid: BINF_PD_CHUNKS=0 / 1 forces the choice);
// beyond 2048 beads it is the only form that keeps the coordinates out of HBM round trips
static bool chunks_pay(int64_t C, int64_t n_beads, int64_t n_pairs)
{
    static std::atomic<int> forced(-2);
    int v = forced.load(std::memory_order_relaxed);
    if (v == -2) {
        const char *e = getenv("BINF_PD_CHUNKS");
        v = e ? (e[0] == '0' ? 0 : 1) : -1;
        forced.store(v, std::memory_order_relaxed);
    }
    if (C < 1 || C > 65535 || n_pairs < 2 * NPY_BUFSIZE || n_pairs > 0x7fffffffLL) return false;
    if (v >= 0) return v == 1;
    return n_beads > 2048 || C < (int64_t)cu_count();
}

static int64_t chi2_chunks(int64_t n_pairs) { return (n_pairs + NPY_BUFSIZE - 1) / NPY_BUFSIZE; }

// the chunk sums of every chain to workspace[0 .. C K), joined into out (log-prob) and / or
// chi2_out (workspace[C K .. C K + C) when the caller passes that)
static int32_t chi2_by_chunks(const PairArgs &a, const RowGeom &g, double *workspace, double *out,
                              double *chi2_out, hipStream_t st)
{
    const int K = (int)chi2_chunks(g.D);
    pairdist_chi2_chunk_kernel<8><<<dim3((unsigned)K, (unsigned)g.C), 256, 0, st>>>(a, g, workspace, K);
    pairdist_chi2_join_kernel<<<dim3((unsigned)((g.C + 255) / 256)), 256, 0, st>>>(g, workspace, K, out, chi2_out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "pairdist chi^2 by chunks");
    return 0;
}

extern "C" int64_t binf_pairdist_chi2_workspace_bytes(int64_t C, int64_t n_beads, int64_t n_pairs)
{
    if (!chunks_pay(C, n_beads, n_pairs)) return 0;
    return (C * chi2_chunks(n_pairs) + C) * (int64_t)sizeof(double);
}

extern "C" int32_t binf_pairdist_gauss_logp_f64(const double *x, const int32_t *pair_i,
                                                const int32_t *pair_j, const double *ys,
                                                double precision,
                                                const double *precision_chain, double *out,
                                                int64_t C, int64_t n_beads,
                                                int64_t n_pairs, void *workspace,
                                                int64_t workspace_bytes, void *stream)
{
    return pairdist_logp_run(x, pair_i, pair_j, ys, precision, precision_chain, out, nullptr, nullptr,
                             C, n_beads, n_pairs, workspace, workspace_bytes, stream);
}

// The same with a per-chain memo of chi^2 (a function of the chain's coordinates alone),
// checked on the device bit for bit: rowsum.hpp, row_memo_check_kernel.
extern "C" int32_t binf_pairdist_gauss_logp_memo_f64(const double *x, const int32_t *pair_i,
                                                     const int32_t *pair_j, const double *ys,
                                                     double precision,
                                                     const double *precision_chain, double *out,
                                                     double *memo_x, double *memo_chi2,
                                                     uint8_t *skip, int64_t C, int64_t n_beads,
                                                     int64_t n_pairs, void *workspace,
                                                     int64_t workspace_bytes, void *stream)
{
    if (C < 0 || n_beads < 0 || n_pairs < 0)
        return fail(BINF_E_ARG, "pairdist_gauss_logp_memo: negative size");
    if (C == 0) return 0;
    if (!x || !memo_x || !memo_chi2 || !skip)
        return fail(BINF_E_ARG, "pairdist_gauss_logp_memo: null buffer");
    // every refusal of the reduction BEFORE the memo is touched (row_memo_check rewrites the
    // stored coordinates of a missed chain; their chi^2 is stored by the reduction after it)
    if (!out || (n_pairs > 0 && (!pair_i || !pair_j || !ys)))
        return fail(BINF_E_ARG, "pairdist_gauss_logp_memo: null buffer");
    int32_t rc = row_reduce_check(C, n_pairs, "pairdist_gauss_logp_memo");
    if (rc) return rc;
    {
        const int64_t need = binf_pairdist_chi2_workspace_bytes(C, n_beads, n_pairs);
        if (need > 0 && workspace && workspace_bytes >= need &&
            (overlap_f64(workspace, need / 8, x, C * 3 * n_beads) || overlap_f64(workspace, need / 8, out, C) ||
             overlap_f64(workspace, need / 8, memo_x, 2 * C * 3 * n_beads) || overlap_f64(workspace, need / 8, memo_chi2, 2 * C)))
            return fail(BINF_E_ALIAS, "pairdist_gauss_logp_memo: the workspace overlaps a buffer");
    }
    rc = row_memo_check(x, memo_x, skip, C, 3 * n_beads, (hipStream_t)stream,
                        "pairdist_gauss_logp_memo check launch");
    if (rc) return rc;
    return pairdist_logp_run(x, pair_i, pair_j, ys, precision, precision_chain, out, skip, memo_chi2,
                             C, n_beads, n_pairs, workspace, workspace_bytes, stream);
}

// tree height of an np.sum over D elements as the block reductions walk it (rowsum.hpp:
// the largest height among the 8192-element chunks)
static int32_t npsum_tree_height(int64_t D)
{
    int32_t H = pairwise_tree_height(D < NPY_BUFSIZE ? D : NPY_BUFSIZE);
    if (D > NPY_BUFSIZE && D % NPY_BUFSIZE != 0) {
        const int32_t h_last = pairwise_tree_height(D % NPY_BUFSIZE);
        if (h_last > H) H = h_last;
    }
    return H;
}

// development aid: BINF_PD_LOGP_LDS_TREE=1 sends the chi^2 of <= 2048 beads through the generic
// block reduction (tree through LDS, two barriers per level) instead of pairdist_chi2_rows_kernel
static bool pairdist_logp_lds_tree()
{
    static std::atomic<int> on(-1);
    int v = on.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("BINF_PD_LOGP_LDS_TREE");
        v = (e && e[0] == '1') ? 1 : 0;
        on.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}

// chains per workgroup of the chi^2 reduction: more than one once there are enough chains
// to fill the chip that way (development aid: BINF_PD_LOGP_ROWS = 1, 2)
static int pairdist_logp_rows(int64_t C, int64_t n_beads, int64_t n_pairs)
{
    static std::atomic<int> forced(-1);
    int v = forced.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("BINF_PD_LOGP_ROWS");
        v = e ? atoi(e) : 0;
        if (v != 1 && v != 2) v = 0;
        forced.store(v, std::memory_order_relaxed);
    }
    int rows = v;
    if (rows == 0) rows = (C >= 1024 && n_pairs >= 2048) ? 2 : 1;
    while (rows > 1 && (int64_t)rows * n_beads * 3 * (int64_t)sizeof(double) > 48 * 1024) rows >>= 1;
    return rows;
}

static int32_t pairdist_logp_run(const double *x, const int32_t *pair_i, const int32_t *pair_j,
                                 const double *ys, double precision, const double *precision_chain,
                                 double *out, const uint8_t *skip, double *memo_chi2, int64_t C,
                                 int64_t n_beads, int64_t n_pairs, void *workspace,
                                 int64_t workspace_bytes, void *stream)
{
    if (C < 0 || n_beads < 0 || n_pairs < 0)
        return fail(BINF_E_ARG, "pairdist_gauss_logp: negative size");
    if (C == 0) return 0;
    if (!out || (n_pairs > 0 && (!x || !pair_i || !pair_j || !ys)))
        return fail(BINF_E_ARG, "pairdist_gauss_logp: null buffer");
    PairArgs a;
    a.x = x; a.I = pair_i; a.J = pair_j; a.ys = ys; a.n_beads = n_beads;
    hipStream_t st = (hipStream_t)stream;
    int32_t rc;
    GaussFinish fin;                // lp = -0.5 chi2 tau + N/2 log tau, written by the reduction
    fin.on = 1; fin.minus = nullptr; fin.tau = precision; fin.tau_chain = precision_chain; fin.n_data = (double)n_pairs;
    {
        // few chains (or more than 2048 beads): every 8192-pair chunk a workgroup of its own
        const int64_t need = binf_pairdist_chi2_workspace_bytes(C, n_beads, n_pairs);
        if (need > 0 && workspace && workspace_bytes >= need) {
            if (overlap_f64(workspace, need / 8, x, C * 3 * n_beads) || overlap_f64(workspace, need / 8, out, C))
                return fail(BINF_E_ALIAS, "pairdist_gauss_logp: the workspace overlaps x or out");
            RowGeom g;
            g.C = C; g.D = (int32_t)n_pairs; g.scale = 1.0; g.fin = fin;
            g.skip = skip; g.way = skip ? skip + C : nullptr; g.memo_sum = memo_chi2;
            g.H = npsum_tree_height(n_pairs);
            if (g.H > 7) return fail(BINF_E_UNSUPPORTED, "pairdist_gauss_logp: pairwise tree height %d", g.H);
            return chi2_by_chunks(a, g, (double *)workspace, out, nullptr, st);
        }
    }
    if (n_beads <= 2048 && !pairdist_logp_lds_tree()) {       // 48 KiB of coordinates fit the LDS budget
        if (C > 0x7fffffffLL || n_pairs > 0x7fffffffLL)
            return fail(BINF_E_UNSUPPORTED, "pairdist_gauss_logp: too large");
        const int rows = pairdist_logp_rows(C, n_beads, n_pairs);
        RowGeom g;
        g.C = C; g.D = (int32_t)n_pairs; g.scale = 1.0; g.fin = fin;
        g.skip = skip; g.way = skip ? skip + C : nullptr; g.memo_sum = memo_chi2;
        g.H = npsum_tree_height(n_pairs);
        if (g.H > 7) return fail(BINF_E_UNSUPPORTED, "pairdist_gauss_logp: pairwise tree height %d", g.H);
        const size_t lds = (size_t)rows * n_beads * 3 * sizeof(double);
        const dim3 grid((unsigned)((C + rows - 1) / rows));
        // two chains per workgroup from 2048 chains up; fewer chains than ~4 workgroups per CU
        // (and rows long enough to feed them): 16 waves per chain instead of 4
        if (rows == 2)
            pairdist_chi2_rows_kernel<2, 8, 256><<<grid, 256, lds, st>>>(a, g, out);
        else if (C < 1024 && n_pairs >= 2048)
            pairdist_chi2_rows_kernel<1, 8, 1024><<<grid, 1024, lds, st>>>(a, g, out);
        else
            pairdist_chi2_rows_kernel<1, 8, 256><<<grid, 256, lds, st>>>(a, g, out);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "pairdist_gauss_logp");
        return 0;
    }
    if (n_beads <= 2048)            // development aid (BINF_PD_LOGP_LDS_TREE=1): the generic block kernel
        rc = row_reduce_launch<PairResidMake, PairArgs, true>(a, C, n_pairs, 1.0, out, st, true,
                                                             "pairdist_gauss_logp",
                                                             (size_t)n_beads * 3 * sizeof(double),
                                                             C < 1024 && n_pairs >= 2048, &fin, skip,
                                                             memo_chi2);
    else
        rc = row_reduce_launch<PairResidMake, PairArgs>(a, C, n_pairs, 1.0, out, st, true,
                                                       "pairdist_gauss_logp", 0, false, &fin, skip,
                                                       memo_chi2);
    if (rc) return rc;
    return 0;
}

extern "C" int32_t binf_pairdist_hmc_energy_f64(const double *x, const double *p,
                                                const int32_t *pair_i, const int32_t *pair_j,
                                                const double *ys, double precision,
                                                const double *precision_chain, double prior_k,
                                                double prior_x0, int32_t n_terms,
                                                const int32_t *term_kind, const double *extra0,
                                                double extra0_scalar, const double *extra1,
                                                double extra1_scalar,
                                                double *energy, double *log_prob, double *memo_x,
                                                double *memo_chi2, uint8_t *memo_state, int64_t C,
                                                int64_t n_beads, int64_t n_pairs, void *workspace,
                                                int64_t workspace_bytes, void *stream)
{
    if (C < 0 || n_beads < 1 || n_pairs < 0)
        return fail(BINF_E_ARG, "pairdist_hmc_energy: bad sizes");
    if (n_terms < 1 || n_terms > 4 || !term_kind)
        return fail(BINF_E_ARG, "pairdist_hmc_energy: 1 to 4 terms, with their kinds");
    int has_prior = 0, seen = 0;
    for (int k = 0; k < n_terms; ++k) {
        const int kind = term_kind[k];
        if (kind < 0 || kind > 3 || (seen & (1 << kind)))
            return fail(BINF_E_ARG, "pairdist_hmc_energy: term kinds are 0..3, each at most once");
        seen |= 1 << kind;
        if (kind == 0) has_prior = 1;
    }
    if (!(seen & 2)) return fail(BINF_E_ARG, "pairdist_hmc_energy: the likelihood (kind 1) must be a term");
    if (C == 0) return 0;
    if (!x || !p || !energy || (n_pairs > 0 && (!pair_i || !pair_j || !ys)))
        return fail(BINF_E_ARG, "pairdist_hmc_energy: null buffer");
    if ((memo_x != nullptr) != (memo_chi2 != nullptr) || (memo_x != nullptr) != (memo_state != nullptr))
        return fail(BINF_E_ARG, "pairdist_hmc_energy: memo_x, memo_chi2 and memo_state go together");
    if (C > 0x7fffffffLL || n_pairs > 0x7fffffffLL)
        return fail(BINF_E_UNSUPPORTED, "pairdist_hmc_energy: too large");
    if (n_beads > 2048)
        return fail(BINF_E_UNSUPPORTED, "pairdist_hmc_energy: n_beads=%lld > 2048 (coordinates staged in LDS)",
                    (long long)n_beads);
    RowGeom g;
    g.C = C; g.D = (int32_t)n_pairs; g.scale = 1.0;
    g.fin.on = 1; g.fin.minus = nullptr; g.fin.tau = precision; g.fin.tau_chain = precision_chain;
    g.fin.n_data = (double)n_pairs;
    g.skip = nullptr; g.way = nullptr; g.memo_sum = memo_chi2;
    g.H = npsum_tree_height(n_pairs);
    PairEnergyArgs a;
    a.x = x; a.p = p; a.I = pair_i; a.J = pair_j; a.ys = ys; a.memo_x = memo_x; a.memo_state = memo_state;
    a.energy = energy; a.log_prob = log_prob;
    a.prior_scale = -0.5 * prior_k; a.prior_x0 = prior_x0;
    a.has_prior = has_prior; a.n_terms = n_terms;
    for (int k = 0; k < 4; ++k) a.term_kind[k] = k < n_terms ? term_kind[k] : 1;
    a.extra[0] = extra0; a.extra[1] = extra1; a.extra_scalar[0] = extra0_scalar; a.extra_scalar[1] = extra1_scalar;
    a.n_beads = (int32_t)n_beads; a.H_d = npsum_tree_height(3 * n_beads);
    a.chi2_in = nullptr;
    if (g.H > 7 || a.H_d > 7)
        return fail(BINF_E_UNSUPPORTED, "pairdist_hmc_energy: pairwise tree height %d", g.H > a.H_d ? g.H : a.H_d);
    hipStream_t st = (hipStream_t)stream;
    {
        // few chains: chi^2 by chunks first (every 8192-pair chunk a workgroup of its own), the rest
        // of the energy from it
        const int64_t need = binf_pairdist_chi2_workspace_bytes(C, n_beads, n_pairs);
        if (need > 0 && workspace && workspace_bytes >= need) {
            if (overlap_f64(workspace, need / 8, x, C * 3 * n_beads) || overlap_f64(workspace, need / 8, p, C * 3 * n_beads) ||
                overlap_f64(workspace, need / 8, energy, C) || (log_prob && overlap_f64(workspace, need / 8, log_prob, C)))
                return fail(BINF_E_ALIAS, "pairdist_hmc_energy: the workspace overlaps a buffer");
            PairArgs pa;
            pa.x = x; pa.I = pair_i; pa.J = pair_j; pa.ys = ys; pa.n_beads = n_beads;
            RowGeom gc = g;
            if (memo_x) {
                // the memo as the log-prob entry point keeps it: checked (and, on a miss, its
                // coordinates rewritten) by a launch of its own, hits skip their chunks, the join
                // stores the new sums -- the same entries and flags the one-launch kernel leaves
                if (overlap_f64(workspace, need / 8, memo_x, 2 * C * 3 * n_beads) || overlap_f64(workspace, need / 8, memo_chi2, 2 * C))
                    return fail(BINF_E_ALIAS, "pairdist_hmc_energy: the workspace overlaps the memo");
                const int32_t rm = row_memo_check(x, memo_x, memo_state, C, 3 * n_beads, st,
                                                  "pairdist_hmc_energy memo check launch");
                if (rm) return rm;
                gc.skip = memo_state; gc.way = memo_state + C; gc.memo_sum = memo_chi2;
            } else {
                gc.memo_sum = nullptr; gc.skip = nullptr; gc.way = nullptr;
            }
            double *chi2 = (double *)workspace + C * chi2_chunks(n_pairs);
            const int32_t rc = chi2_by_chunks(pa, gc, (double *)workspace, nullptr, chi2, st);
            if (rc) return rc;
            a.chi2_in = chi2;
        }
    }
    const int rows = pairdist_logp_rows(C, n_beads, n_pairs);
    const size_t lds = (size_t)rows * n_beads * 3 * sizeof(double);
    if (rows == 2)
        pairdist_energy_kernel<2, 8, 256><<<dim3((unsigned)((C + 1) / 2)), 256, lds, st>>>(a, g);
    else if (C < 1024 && n_pairs >= 2048)           // few chains: 16 waves per chain (as the log-prob)
        pairdist_energy_kernel<1, 8, 1024><<<dim3((unsigned)C), 1024, lds, st>>>(a, g);
    else
        pairdist_energy_kernel<1, 8, 256><<<dim3((unsigned)C), 256, lds, st>>>(a, g);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "pairdist_hmc_energy launch");
    return 0;
}

static bool sym_serves(int64_t n_beads)
{
    return n_beads >= SYM_MIN_BEADS && n_beads <= SYM_MAX_BEADS && sym_enabled();
}

// 257 .. 1024 beads WITH packed targets: the ring kernels (development aid: BINF_PD_RING=0
// keeps the one-sided loops)
static bool ring_serves(int64_t n_beads)
{
    static std::atomic<int> on(-1);
    int v = on.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("BINF_PD_RING");
        v = (e && e[0] == '0') ? 0 : 1;
        on.store(v, std::memory_order_relaxed);
    }
    return v != 0 && n_beads > SYM_MAX_BEADS && n_beads <= RING_MAX_BEADS;
}

// ... and up to TILES_MAX_BEADS the tile kernels alone (no workgroup-per-chain form there)
static bool tiles_serve(int64_t n_beads)
{
    return ring_serves(n_beads) || (ring_serves(RING_MAX_BEADS) && n_beads > RING_MAX_BEADS && n_beads <= TILES_MAX_BEADS);
}

// workgroups of the ring kernels: as many as the chip holds at a time (NBLK <= 8: two
// per CU, 16 waves; else one), each walks its share of the chains
static unsigned ring_grid(int64_t C, int nblk)
{
    const int64_t slots = (int64_t)cu_count() * (nblk <= 8 ? 2 : 1);
    return (unsigned)(C < slots ? C : slots);
}

// Few chains of 257..1024 beads: every tile a wave of its own (pairdist_tiles_kernel) when the
// caller brings the workspace for the tiles' partial sums -- same bits as the ring kernels,
// which serve every other case.  A workgroup per chain keeps 64 NBLK threads busy per chain:
// tiles pay while that leaves most of the chip idle (development aid: BINF_PD_TILES=0 / 1
// forces the choice).
static bool tiles_pay(int64_t C, int64_t n_beads)
{
    static std::atomic<int> forced(-2);
    int v = forced.load(std::memory_order_relaxed);
    if (v == -2) {
        const char *e = getenv("BINF_PD_TILES");
        v = e ? (e[0] == '0' ? 0 : 1) : -1;
        forced.store(v, std::memory_order_relaxed);
    }
    if (!tiles_serve(n_beads) || C > 65535) return false;
    if (n_beads > RING_MAX_BEADS) return true;       // the only symmetric form beyond 1024 beads
    if (v >= 0) return v == 1;
    // measured cross-over (scripts/probe_pairdist_few_chains.py, 256 CUs): a sample() of L = 20
    // costs 0.52 / 0.87 / 2.9 ms with a workgroup per chain at 320 / 512 / 1024 beads whatever
    // the number of chains up to ~256, and 0.35 + 0.0013 C / 0.43 + 0.0033 C / 0.82 + 0.0105 C ms
    // with a wave per tile: tiles win up to ~140 / 130 / 195 chains
    const int64_t nblk = (n_beads + 63) / 64;
    return C <= (96 + 6 * nblk) * (int64_t)cu_count() / 256;
}

static void launch_tiles(const double *x, const double *ypk, double *part, int64_t C, int64_t n, hipStream_t st)
{
    const int nblk = (int)((n + 63) / 64);
    const dim3 grid((unsigned)((ring_tiles(nblk) + TILE_WAVES - 1) / TILE_WAVES), (unsigned)C);
    if (n == 64 * nblk) pairdist_tiles_kernel<true><<<grid, 64 * TILE_WAVES, 0, st>>>(x, ypk, part, (int32_t)n, nblk);
    else                pairdist_tiles_kernel<false><<<grid, 64 * TILE_WAVES, 0, st>>>(x, ypk, part, (int32_t)n, nblk);
}

static void launch_grad_ring(const double *x, const double *ypk, double precision,
                             const double *precision_chain, double *out, int64_t C, int64_t n,
                             hipStream_t st)
{
    const int nblk = (int)((n + 63) / 64);
    const dim3 grid(ring_grid(C, nblk)), block(64 * nblk);
    const bool full = n == 64 * nblk;
#define GRAD_RING(FULLV, NBTV) \
    pairdist_grad_ring_kernel<FULLV, NBTV><<<grid, block, 0, st>>>(x, ypk, precision, precision_chain, out, \
                                                                   (int32_t)n, nblk, C)
    if (nblk <= 8) { if (full) GRAD_RING(true, 8); else GRAD_RING(false, 8); }
    else           { if (full) GRAD_RING(true, 16); else GRAD_RING(false, 16); }
#undef GRAD_RING
}

static void launch_leapfrog_ring(const PairLeapArgs &a, bool fma, hipStream_t st)
{
    const int nblk = (a.n_beads + 63) / 64;
    const dim3 grid(ring_grid(a.n_chains, nblk)), block(64 * nblk);
    const bool full = a.n_beads == 64 * nblk;
#define LEAP_RING(FMAV, FULLV) \
    do { \
        if (nblk <= 8) pairdist_leapfrog_ring_kernel<FMAV, FULLV, 8><<<grid, block, 0, st>>>(a, nblk); \
        else           pairdist_leapfrog_ring_kernel<FMAV, FULLV, 16><<<grid, block, 0, st>>>(a, nblk); \
    } while (0)
    if (fma) { if (full) LEAP_RING(true, true); else LEAP_RING(true, false); }
    else     { if (full) LEAP_RING(false, true); else LEAP_RING(false, false); }
#undef LEAP_RING
}

extern "C" int64_t binf_pairdist_tiles_workspace_bytes(int64_t C, int64_t n_beads)
{
    if (C < 1 || n_beads < 1 || !tiles_pay(C, n_beads)) return 0;
    const int64_t nblk = (n_beads + 63) / 64;
    return C * ring_tiles((int)nblk) * 2 * 3 * 64 * (int64_t)sizeof(double);
}

extern "C" int64_t binf_pairdist_packed_targets_bytes(int64_t n_beads)
{
    const int64_t nblk = (n_beads + 63) / 64;
    if (tiles_serve(n_beads)) return nblk * ring_steps((int)nblk) * 64 * (int64_t)sizeof(double);
    if (!sym_serves(n_beads)) return 0;
    return nblk * nblk * SYM_STEPS * 64 * (int64_t)sizeof(double);
}

extern "C" int32_t binf_pairdist_pack_targets_f64(const double *ymat, double *packed,
                                                  int64_t n_beads, void *stream)
{
    if (n_beads < 1) return fail(BINF_E_ARG, "pairdist_pack_targets: bad size");
    if (!sym_serves(n_beads) && !tiles_serve(n_beads))
        return fail(BINF_E_UNSUPPORTED, "pairdist_pack_targets: n_beads=%lld has no packed form "
                    "(binf_pairdist_packed_targets_bytes is 0)", (long long)n_beads);
    if (!ymat || !packed) return fail(BINF_E_ARG, "pairdist_pack_targets: null buffer");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (int)((n_beads + 63) / 64);
    if (tiles_serve(n_beads)) {
        ring_pack_targets_kernel<<<dim3((unsigned)nblk), 256, 0, st>>>(ymat, packed, (int)n_beads, nblk);
        const hipError_t er = hipGetLastError();
        if (er != hipSuccess) return hip_fail(er, "pairdist_pack_targets launch");
        return 0;
    }
    const dim3 block(64 * nblk * nblk);
    if (nblk == 1)      sym_pack_targets_kernel<1><<<1, block, 0, st>>>(ymat, packed, (int)n_beads);
    else if (nblk == 2) sym_pack_targets_kernel<2><<<1, block, 0, st>>>(ymat, packed, (int)n_beads);
    else if (nblk == 3) sym_pack_targets_kernel<3><<<1, block, 0, st>>>(ymat, packed, (int)n_beads);
    else                sym_pack_targets_kernel<4><<<1, block, 0, st>>>(ymat, packed, (int)n_beads);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "pairdist_pack_targets launch");
    return 0;
}

extern "C" int32_t binf_pairdist_gauss_grad_f64(const double *x, const double *ymat,
                                                double precision,
                                                const double *precision_chain,
                                                double *out, int64_t C,
                                                int64_t n_beads, void *stream)
{
    return binf_pairdist_gauss_grad_packed_f64(x, ymat, nullptr, precision, precision_chain, out, C,
                                               n_beads, nullptr, 0, stream);
}

extern "C" int32_t binf_pairdist_gauss_grad_packed_f64(const double *x, const double *ymat,
                                                       const double *packed, double precision,
                                                       const double *precision_chain,
                                                       double *out, int64_t C,
                                                       int64_t n_beads, void *workspace,
                                                       int64_t workspace_bytes, void *stream)
{
    if (C < 0 || n_beads < 1) return fail(BINF_E_ARG, "pairdist_gauss_grad: bad sizes");
    if (C == 0) return 0;
    if (!x || !ymat || !out) return fail(BINF_E_ARG, "pairdist_gauss_grad: null buffer");
    if (n_beads > 46340 || C > 0x7fffffffLL)
        return fail(BINF_E_UNSUPPORTED, "pairdist_gauss_grad: too large");
    const size_t lds = (size_t)n_beads * 3 * sizeof(double);
    hipStream_t gst = (hipStream_t)stream;
    if (sym_serves(n_beads)) {
        const int nblk = (int)((n_beads + 63) / 64);
        if (nblk == 1)      launch_grad_sym<1>(x, ymat, packed, precision, precision_chain, out, C, n_beads, gst);
        else if (nblk == 2) launch_grad_sym<2>(x, ymat, packed, precision, precision_chain, out, C, n_beads, gst);
        else if (nblk == 3) launch_grad_sym<3>(x, ymat, packed, precision, precision_chain, out, C, n_beads, gst);
        else                launch_grad_sym<4>(x, ymat, packed, precision, precision_chain, out, C, n_beads, gst);
    }
    else if (packed && tiles_serve(n_beads) &&
             (ring_serves(n_beads) || (workspace && workspace_bytes >= binf_pairdist_tiles_workspace_bytes(C, n_beads) &&
                                       binf_pairdist_tiles_workspace_bytes(C, n_beads) > 0))) {
        const int64_t need = binf_pairdist_tiles_workspace_bytes(C, n_beads);
        if (need > 0 && workspace && workspace_bytes >= need) {
            if (overlap_f64(workspace, need / 8, x, C * 3 * n_beads) || overlap_f64(workspace, need / 8, out, C * 3 * n_beads))
                return fail(BINF_E_ALIAS, "pairdist_gauss_grad: the workspace overlaps x or out");
            launch_tiles(x, packed, (double *)workspace, C, n_beads, gst);
            pairdist_tiles_grad_kernel<<<dim3((unsigned)((n_beads + 255) / 256), (unsigned)C), 256, 0, gst>>>(
                (const double *)workspace, precision, precision_chain, out, (int32_t)n_beads,
                (int32_t)((n_beads + 63) / 64));
        } else {
            launch_grad_ring(x, packed, precision, precision_chain, out, C, n_beads, gst);
        }
    }
    else if (n_beads <= 1024 && lanes_per_bead(C) == 4)
        pairdist_grad4_kernel<<<dim3((unsigned)C), 1024, lds, (hipStream_t)stream>>>(
            x, ymat, precision, precision_chain, out, (int32_t)n_beads);
    else if (n_beads <= 1024)
        pairdist_grad1_kernel<<<dim3((unsigned)C), 256, lds, (hipStream_t)stream>>>(
            x, ymat, precision, precision_chain, out, (int32_t)n_beads);
    else
        pairdist_grad_kernel<256><<<dim3((unsigned)C), 256, 0, (hipStream_t)stream>>>(
            x, ymat, precision, precision_chain, out, (int32_t)n_beads);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "pairdist_gauss_grad launch");
    return 0;
}

extern "C" int32_t binf_pairdist_leapfrog_f64(double *q, double *p, const double *ymat,
                                              double precision, const double *precision_chain,
                                              int32_t has_prior, double prior_k, double prior_x0,
                                              int32_t prior_first, double timestep,
                                              const double *dt_chain, int32_t nsteps, int64_t C,
                                              int64_t n_beads, int32_t mode, void *stream)
{
    return binf_pairdist_leapfrog_packed_f64(q, nullptr, p, ymat, nullptr, precision, precision_chain,
                                             has_prior, prior_k, prior_x0, prior_first, timestep, dt_chain,
                                             nsteps, C, n_beads, mode, nullptr, 0, stream);
}

extern "C" int32_t binf_pairdist_leapfrog_packed_f64(double *q, const double *q_from, double *p,
                                                     const double *ymat,
                                                     const double *packed, double precision,
                                                     const double *precision_chain,
                                                     int32_t has_prior, double prior_k, double prior_x0,
                                                     int32_t prior_first, double timestep,
                                                     const double *dt_chain, int32_t nsteps, int64_t C,
                                                     int64_t n_beads, int32_t mode, void *workspace,
                                                     int64_t workspace_bytes, void *stream)
{
    if (C < 0 || n_beads < 1 || nsteps < 1)
        return fail(BINF_E_ARG, "pairdist_leapfrog: need C>=0, n_beads>=1, nsteps>=1");
    if (mode != BINF_MODE_EXACT && mode != BINF_MODE_FMA)
        return fail(BINF_E_ARG, "pairdist_leapfrog: unknown mode %d", mode);
    if (C == 0) return 0;
    if (!q || !p || !ymat) return fail(BINF_E_ARG, "pairdist_leapfrog: null buffer");
    const bool beyond = n_beads > 1024;          // only as a wave per tile (packed targets + workspace)
    if (beyond && !(packed && tiles_serve(n_beads) && workspace &&
                    binf_pairdist_tiles_workspace_bytes(C, n_beads) > 0 &&
                    workspace_bytes >= binf_pairdist_tiles_workspace_bytes(C, n_beads)))
        return fail(BINF_E_UNSUPPORTED, "pairdist_leapfrog: n_beads=%lld > 1024 needs packed targets and the "
                    "workspace of binf_pairdist_tiles_workspace_bytes (up to %d beads)", (long long)n_beads,
                    TILES_MAX_BEADS);
    if (C > 0x7fffffffLL) return fail(BINF_E_UNSUPPORTED, "pairdist_leapfrog: too many chains");
    // q_from may be q itself (in place) or a separate buffer; a PARTIAL overlap -- with q, or with
    // the momenta the launch updates -- would have chains read what others already wrote
    if (q_from && q_from != q && (overlap_f64(q_from, C * 3 * n_beads, q, C * 3 * n_beads) ||
                                  overlap_f64(q_from, C * 3 * n_beads, p, C * 3 * n_beads)))
        return fail(BINF_E_ALIAS, "pairdist_leapfrog: q_from may be exactly q or a separate buffer, "
                    "not a partial overlap with q or p");
    if (overlap_f64(q, C * 3 * n_beads, p, C * 3 * n_beads))
        return fail(BINF_E_ALIAS, "pairdist_leapfrog: q and p overlap");
    PairLeapArgs a;
    a.q = q; a.q_from = (q_from == q) ? nullptr : q_from; a.p = p; a.ymat = ymat; a.ypk = packed;
    a.tau_chain = precision_chain; a.dt_chain = dt_chain;
    a.tau = precision; a.timestep = timestep; a.prior_k = prior_k; a.prior_x0 = prior_x0;
    a.has_prior = has_prior ? 1 : 0; a.prior_first = prior_first ? 1 : 0;
    a.nsteps = nsteps; a.n_beads = (int32_t)n_beads; a.n_chains = C;
    dim3 grid((unsigned)C);
    hipStream_t st = (hipStream_t)stream;
    const bool fma = mode == BINF_MODE_FMA;
    const int nb = (int)((n_beads + 255) / 256);
    const bool four = lanes_per_bead(C) == 4;
    // positions; four tiles x four lanes keep the momentum there as well
    const size_t lds = (size_t)n_beads * 3 * sizeof(double) * ((nb >= 4 && four) ? 2 : 1);
    if (sym_serves(n_beads)) {
        const int nblk = (int)((n_beads + 63) / 64);
        if (nblk == 1)      launch_leapfrog_sym<1>(a, fma, st);
        else if (nblk == 2) launch_leapfrog_sym<2>(a, fma, st);
        else if (nblk == 3) launch_leapfrog_sym<3>(a, fma, st);
        else                launch_leapfrog_sym<4>(a, fma, st);
        hipError_t es = hipGetLastError();
        if (es != hipSuccess) return hip_fail(es, "pairdist_leapfrog launch");
        return 0;
    }
    if (packed && tiles_serve(n_beads)) {
        const int64_t need = binf_pairdist_tiles_workspace_bytes(C, n_beads);
        if (need > 0 && workspace && workspace_bytes >= need) {
            const int64_t nq = C * 3 * n_beads;
            if (overlap_f64(workspace, need / 8, q, nq) || overlap_f64(workspace, need / 8, p, nq) ||
                (q_from && overlap_f64(workspace, need / 8, q_from, nq)))
                return fail(BINF_E_ALIAS, "pairdist_leapfrog: the workspace overlaps q, q_from or p");
            // few chains: per force evaluation one launch of a wave per tile and one that adds the
            // partial sums and kicks / drifts -- the ring kernel's arithmetic, bit for bit
            const int nblk = (int)((n_beads + 63) / 64);
            const dim3 ug((unsigned)((n_beads + 255) / 256), (unsigned)C);
            for (int e = 0; e <= nsteps; ++e) {
                const double *qin = (e == 0 && a.q_from) ? a.q_from : q;
                launch_tiles(qin, packed, (double *)workspace, C, n_beads, st);
                if (fma) pairdist_tiles_update_kernel<true><<<ug, 256, 0, st>>>((const double *)workspace, a, qin, nblk, e);
                else     pairdist_tiles_update_kernel<false><<<ug, 256, 0, st>>>((const double *)workspace, a, qin, nblk, e);
            }
        } else if (ring_serves(n_beads)) {
            launch_leapfrog_ring(a, fma, st);
        } else {
            return fail(BINF_E_UNSUPPORTED, "pairdist_leapfrog: n_beads=%lld needs the tiles workspace", (long long)n_beads);
        }
        hipError_t es = hipGetLastError();
        if (es != hipSuccess) return hip_fail(es, "pairdist_leapfrog launch");
        return 0;
    }
#define LAUNCH(NBV)                                                                         \
    do {                                                                                    \
        if (four) {                                                                         \
            if (fma) pairdist_leapfrog_kernel<NBV, true, 4><<<grid, 1024, lds, st>>>(a);    \
            else     pairdist_leapfrog_kernel<NBV, false, 4><<<grid, 1024, lds, st>>>(a);   \
        } else {                                                                            \
            if (fma) pairdist_leapfrog_kernel<NBV, true, 1><<<grid, 256, lds, st>>>(a);     \
            else     pairdist_leapfrog_kernel<NBV, false, 1><<<grid, 256, lds, st>>>(a);    \
        }                                                                                   \
    } while (0)
    if (nb <= 1) LAUNCH(1);
    else if (nb == 2) LAUNCH(2);
    else if (nb == 3) LAUNCH(3);
    else LAUNCH(4);
#undef LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "pairdist_leapfrog launch");
    return 0;
}

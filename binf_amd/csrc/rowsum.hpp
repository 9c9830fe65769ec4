// Row reductions in numpy's summation order, generic over the element
// function: out[row] = scale * np.sum([f(row, i) for i in range(D)]).
// Shared by the generic tier (sum / sum of squares) and the polynomial model
// (squared residuals evaluated on the fly).  gfx950, wave64.
#pragma once
#include "common.hpp"

namespace binf {

constexpr int NPY_BUFSIZE = 8192;   // numpy's ufunc buffer, in elements

// Sum of one pairwise leaf (<=128 elements starting at `off`) by the 8 lanes
// of a lane group; every lane of the wave must call it (shuffles).  Returns
// the leaf sum in all 8 lanes.  f(i) = value of element i of the row.
template <class F>
__device__ inline double leaf_sum_f(const F &f, int off, int n, int lane,
                                    bool active)
{
    const int j = lane & 7;
    const int T = (n >= 8) ? (n >> 3) : 0;
    const int rem = (n >= 8) ? (n & 7) : n;
    double r = 0.0;
    if (active && T > 0) {
        // eight element values at a time: their loads (and the gathers behind them) are
        // in flight together, then they are added in order -- one value per round trip
        // left a lane waiting for memory 16 times per leaf
        int t = 0;
        for (; t + 8 <= T; t += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = f(off + 8 * (t + u) + j);
            r = (t == 0) ? v[0] : r + v[0];
#pragma unroll
            for (int u = 1; u < 8; ++u) r = r + v[u];
        }
        for (; t < T; ++t) {
            const double v = f(off + 8 * t + j);
            r = (t == 0) ? v : r + v;
        }
    }
    r = sum8_f64(r);
    double res = (T > 0) ? r : -0.0;
    // the leaf's tail elements (n not a multiple of 8), in order; skipped when no leaf
    // of this wave has one
    if (__any(rem != 0)) {
        double tail = 0.0;
        if (active && j < rem) tail = f(off + 8 * T + j);
        const int leafbase = lane & ~7;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const double v = shfl_f64(tail, leafbase + i);
            const double s = res + v;
            res = (i < rem) ? s : res;
        }
    }
    return res;
}

// Optional epilogue of a chi^2 reduction: the Gaussian error model's log-prob
//   lp[c] = -0.5 * chi2[c] * tau_c + N * 0.5 * log(tau_c)         likelihood.py:54-57
// written by the reduction itself instead of a second launch.
struct GaussFinish {
    int32_t on;              // 1: the log-prob above; 2: scale * sum - minus[row]
    double tau;
    const double *tau_chain;
    double n_data;
    const double *minus;     // on == 2 (HMC energy: 0.5 sum p^2 - log_prob, hmc.py:148)
};

struct RowGeom {
    int64_t C;
    int32_t D;
    int32_t H;       // largest pairwise tree height among the row's 8192-element chunks
    double scale;
    GaussFinish fin;
    // per-row memo of the raw sum (block kernel only), two entries per row: a row whose skip
    // flag is set takes memo_sum[way[row]][row] instead of being summed again; every row
    // summed leaves its sum there (row_memo_check_kernel sets skip and way)
    const uint8_t *skip;     // [C]
    const uint8_t *way;      // [C]
    double *memo_sum;        // [2][C]
};

__device__ inline double *row_memo_slot(const RowGeom &g, int64_t row)
{
    return g.memo_sum + (int64_t)g.way[row] * g.C + row;
}

__device__ inline double row_result(const RowGeom &g, int64_t row, double sum)
{
    const double v = g.scale * sum;
    if (!g.fin.on) return v;
    if (g.fin.on == 2) return v - g.fin.minus[row];
    const double t = g.fin.tau_chain ? g.fin.tau_chain[row] : g.fin.tau;
    const double logZ = g.fin.n_data * 0.5 * log(t);
    return -0.5 * v * t + logZ;
}

// FM: functor factory -- FM::make(args, row) returns the per-row element
// functor; ARGS is passed by value as kernel argument.

// H <= 3 and D <= 8192: G = 8<<H lanes of one wave per row.
template <class FM, class ARGS>
__global__ void __launch_bounds__(256)
row_reduce_wave_kernel(const ARGS args, const RowGeom g, double *out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int H = g.H;
    const int lg = 3 + H;
    const int slot = lane & ((1 << lg) - 1);
    const int64_t row_raw = (wave << (6 - lg)) + (lane >> lg);
    const bool valid = row_raw < g.C;
    const int64_t row = valid ? row_raw : g.C - 1;
    const Leaf L = pairwise_leaf(g.D, H, slot >> 3);
    const auto f = FM::make(args, row);
    double res = leaf_sum_f(f, L.off, L.len, lane, true);
    for (int l = 0; l < H; ++l) {
        const double o = xor_level_f64(res, l, lane);
        const double s = res + o;
        res = (L.depth >= H - l) ? s : res;
    }
    if (valid && slot == 0) out[row] = row_result(g, row, 0.0 + res);
}

// Any D: one 256-thread workgroup per row.  numpy's buffered reduction feeds
// the pairwise loop NPY_BUFSIZE elements at a time and adds the chunk sums up
// one after the other.  A chunk's tree has height <= 7: 6 for a full chunk,
// 7 for the 441 ragged lengths in [7689, 8191].  g.H is the largest height
// among the row's chunks; walking a shallower chunk with it only adds
// redundant paths (pairwise_leaf).
// STAGED: the functor factory first copies per-row data into dynamic LDS
// (FM::stage, all 256 threads) and builds the element functor on top of it
// (FM::make_lds) -- for element functions that gather from a small per-row
// table (pair distances: the chain's coordinates).
// THREADS: 256, or 1024 when a launch has too few rows to fill the chip with
// 4-wave workgroups (16 waves per row hide the latency of the element function).
template <class FM, class ARGS, bool STAGED = false, int THREADS = 256>
__global__ void __launch_bounds__(THREADS)
row_reduce_block_kernel(const ARGS args, const RowGeom g, double *out)
{
    constexpr int GROUPS = THREADS / 8;
    __shared__ double S[128];
    __shared__ int dep[128];
    extern __shared__ double row_lds[];
    const int H = g.H;
    const int npaths = 1 << H;
    const int lane = threadIdx.x & 63;
    const int group = threadIdx.x >> 3;      // GROUPS groups of 8 lanes
    const int64_t row = blockIdx.x;
    if (g.skip && g.skip[row]) {                 // workgroup-uniform
        if (threadIdx.x == 0) out[row] = row_result(g, row, *row_memo_slot(g, row));
        return;
    }
    if constexpr (STAGED) {
        FM::stage(args, row, row_lds);
        __syncthreads();
    }
    const auto f = [&]() {
        if constexpr (STAGED) return FM::make_lds(args, row, row_lds);
        else return FM::make(args, row);
    }();
    double total = 0.0;                      // the reduction's identity
    for (int cbase = 0; cbase == 0 || cbase < g.D; cbase += NPY_BUFSIZE) {
        const int n = (g.D - cbase < NPY_BUFSIZE) ? g.D - cbase : NPY_BUFSIZE;
        for (int base = 0; base < npaths; base += GROUPS) {
            const int path = base + group;
            const bool act = path < npaths;
            const Leaf L = pairwise_leaf(n, H, act ? path : 0);
            const double s = leaf_sum_f(f, cbase + L.off, L.len, lane, act);
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
    if (threadIdx.x == 0) {
        if (g.memo_sum) *row_memo_slot(g, row) = total;
        out[row] = row_result(g, row, total);
    }
}

// The shapes row_reduce_launch accepts, as a check of its own: a caller that changes
// state BEFORE the reduction (the chi^2 memos: row_memo_check rewrites the memo's
// arguments) must know the reduction will not be refused afterwards -- an entry whose
// arguments match but whose sum was never stored would be a silent wrong hit.
static int32_t row_reduce_check(int64_t C, int64_t D, const char *what, int32_t *height = nullptr)
{
    if (C > 0x7fffffffLL || D > 0x7fffffffLL)
        return fail(BINF_E_UNSUPPORTED, "%s: too large", what);
    int32_t H = pairwise_tree_height(D < NPY_BUFSIZE ? D : NPY_BUFSIZE);
    if (D > NPY_BUFSIZE && D % NPY_BUFSIZE != 0) {
        const int32_t h_last = pairwise_tree_height(D % NPY_BUFSIZE);
        if (h_last > H) H = h_last;
    }
    if (H > 7) return fail(BINF_E_UNSUPPORTED, "%s: pairwise tree height %d", what, H);
    if (height) *height = H;
    return 0;
}

template <class FM, class ARGS, bool STAGED = false>
static int32_t row_reduce_launch(const ARGS &args, int64_t C, int64_t D,
                                 double scale, double *out, hipStream_t st,
                                 bool force_block, const char *what,
                                 size_t staged_bytes = 0, bool wide = false,
                                 const GaussFinish *fin = nullptr,
                                 const uint8_t *memo_state = nullptr, double *memo_sum = nullptr)
{
    int32_t height = 0;
    const int32_t rc_shape = row_reduce_check(C, D, what, &height);
    if (rc_shape) return rc_shape;
    RowGeom g;
    g.C = C; g.D = (int32_t)D; g.scale = scale;
    if (fin) g.fin = *fin;
    else { g.fin.on = 0; g.fin.tau = 1.0; g.fin.tau_chain = nullptr; g.fin.n_data = 0.0; g.fin.minus = nullptr; }
    // honoured by the block kernel (force_block); memo_state = [skip [C], way [C]]
    g.skip = memo_state; g.way = memo_state ? memo_state + C : nullptr; g.memo_sum = memo_sum;
    g.H = height;
    if constexpr (STAGED) {
        if (wide)
            row_reduce_block_kernel<FM, ARGS, true, 1024><<<dim3((unsigned)C), 1024, staged_bytes, st>>>(args, g, out);
        else
            row_reduce_block_kernel<FM, ARGS, true><<<dim3((unsigned)C), 256, staged_bytes, st>>>(args, g, out);
    } else if (g.H <= 3 && !force_block) {
        const int64_t rows_per_wave = 64 >> (3 + g.H);
        const int64_t waves = (C + rows_per_wave - 1) / rows_per_wave;
        const int64_t blocks = (waves + 3) / 4;
        row_reduce_wave_kernel<FM, ARGS><<<dim3((unsigned)blocks), 256, 0, st>>>(args, g, out);
    } else {
        row_reduce_block_kernel<FM, ARGS><<<dim3((unsigned)C), 256, 0, st>>>(args, g, out);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what);
    return 0;
}

// ---- per-row memo of a reduction whose value depends on a small per-row argument vector
// (the chi^2 of a likelihood as a function of a chain's coefficients / coordinates).
// TWO entries per row: HMCSampler.sample() evaluates the state (E_before, hmc.py:148) and
// the proposal (E_after, hmc.py:150), and the next call's state is one of the two --
// whichever way the acceptance test went, it is in the memo.
// One wave per row: lane k compares argument k BIT FOR BIT with both entries' copies.  A
// row that equals one gets skip = 1 and way = that entry; a row that equals neither gets
// skip = 0, way = the entry NOT used last, and its arguments copied there (the block
// reduction that follows stores the new sum in memo_sum[way]).  Content-checked on the
// device: no tensor identities, versions or host synchronisation involved.
// memo_arg [2][C][K], state [2][C] = skip flags, then ways.
template <int UNUSED = 0>
__global__ void __launch_bounds__(256)
row_memo_check_kernel(const double *arg, double *memo_arg, uint8_t *state, int64_t C, int32_t K)
{
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    const double *a = arg + c * K;
    double *m0 = memo_arg + c * K, *m1 = memo_arg + (C + c) * K;
    bool same0 = true, same1 = true;
    for (int k = lane; k < K; k += 64) {
        const long long bits = __double_as_longlong(a[k]);
        same0 = same0 && (bits == __double_as_longlong(m0[k]));
        same1 = same1 && (bits == __double_as_longlong(m1[k]));
    }
    const bool hit0 = __all(same0), hit1 = __all(same1);
    uint8_t *skip = state, *way = state + C;
    int w;
    if (hit0 || hit1) {
        w = hit0 ? 0 : 1;
    } else {
        w = 1 - (way[c] & 1);
        double *m = w ? m1 : m0;
        for (int k = lane; k < K; k += 64) m[k] = a[k];
    }
    if (lane == 0) {
        skip[c] = (hit0 || hit1) ? 1 : 0;
        way[c] = (uint8_t)w;
    }
}

static int32_t row_memo_check(const double *arg, double *memo_arg, uint8_t *state, int64_t C,
                              int64_t K, hipStream_t st, const char *what)
{
    if (C > 0x7fffffffLL * 4 || K > 0x7fffffffLL)
        return fail(BINF_E_UNSUPPORTED, "%s: too large", what);
    row_memo_check_kernel<><<<dim3((unsigned)((C + 3) / 4)), 256, 0, st>>>(arg, memo_arg, state, C, (int32_t)K);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what);
    return 0;
}

// Gaussian error model on top of a chi^2 row reduction (shared by the polynomial
// and the pair-distance likelihoods)
// lp[c] = -0.5 * chi2[c] * tau_c + N * 0.5 * log(tau_c)     likelihood.py:54-57
static __global__ void gauss_logp_finish_kernel(const double *chi2, double tau,
                                         const double *tau_chain, double *out,
                                         int64_t C, double n_data)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double t = tau_chain ? tau_chain[c] : tau;
    const double logZ = n_data * 0.5 * log(t);
    out[c] = -0.5 * chi2[c] * t + logZ;
}

}  // namespace binf

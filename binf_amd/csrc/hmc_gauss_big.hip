// Fused Gaussian HMC for chains of ANY length (D > 8192, or a length <= 8192 whose
// pairwise tree is deeper than the persistent kernel handles): one
// HMCSampler.sample() (binf/samplers/hmc.py:136-164,183-191) on the reference's
// TestHO Gaussian (binf/pdf/__init__.py:181-191) in three launches instead of
// ~3 L launches of the per-step tier, 24 D bytes per chain instead of 32 D per
// leapfrog step.  Same arithmetic, same bits as the other tiers.  gfx950, wave64.
//
// np.sum of a long vector is a CHAIN of chunk sums: numpy's buffered reduction
// feeds its pairwise loop 8192 elements at a time, r = r + pairwise(chunk).  So:
//
//  1. trajectory kernel -- one 256-thread workgroup per (chain, 8192-chunk).
//     A group of 8 lanes owns one leaf of the chunk's pairwise tree (<= 128
//     elements, lane j the elements off + 8 t + j -- numpy's j-th accumulator);
//     the trajectory is elementwise, so q and p (16 + 16 doubles per lane) stay
//     in registers for all L steps; the proposal goes straight to q_out.  The
//     four sums the energies need (q0^2, p0^2, q_L^2, p_L^2) are formed per lane,
//     per leaf (xor-shuffles 1, 2, 4 + tail elements), then up the chunk's tree
//     through LDS -- bit-identical to pairwise(chunk) -- and written to a small
//     workspace [C][chunks][4].
//  2. finish kernel -- one thread per chain adds the chunk sums in order, forms
//     E_before / E_after, runs the clipped-exp Metropolis test, adapts, counts.
//  3. restore kernel -- flat 16-byte copy of q0 over q_out for REJECTED chains
//     only (accepted chains cost one flag read per 2 KiB).
#include "gauss_common.hpp"
#include "rowsum.hpp"
#include "xoshiro.hpp"

namespace binf {

struct BigArgs {
    const double *q0;
    const double *p0;
    double *q_out;
    double *ws;              // [C][nchunks][4]: sum q0^2, p0^2, qL^2, pL^2 per chunk
    const double *dt_chain;
    double timestep;
    double k;
    double x0;
    int64_t C;
    int64_t D;
    int32_t nchunks;
    int32_t nsteps;
    int32_t H;               // tree height of the chunks this launch covers (<= 7)
    int32_t chunk0;          // first chunk of this launch
    int32_t chunks_here;     // chunks per chain in this launch
    // draws generated in the kernel (RNG != 0)
    uint64_t rng_seed;
    uint64_t rng_offset;
    int64_t chain_offset;    // global index of this launch's first chain (sharded runs)
    double *p_dump;          // [C x D], RNG == 2 only
};

// RNG = 0: the momentum is read from HBM; 1: generated in the kernel (xoshiro.hpp),
// no momentum buffer; 2: only written out (p_dump), nothing integrated -- the
// handle by which the in-kernel generator is tested (fused == sampling from its dump).
enum { BIG_RNG_HBM = 0, BIG_RNG_FUSED = 1, BIG_RNG_DUMP = 2 };
// the stream of the chain's acceptance draw (finish kernel): its own id space
constexpr uint64_t BIG_U_STREAM = 1ull << 62;

// Leaf sum of per-lane register values: in-lane running sum r over t < T
// (numpy's accumulator j), combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then
// the leaf's tail elements in order.  All lanes of the wave call it.
template <bool REG>
__device__ inline double leaf_finish(double r, double tail, int T, int rem, int lane)
{
    r = sum8_f64(r);
    if (REG) return r;                       // 128 elements: no tail
    double res = (T > 0) ? r : -0.0;
    const int leafbase = lane & ~7;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const double v = shfl_f64(tail, leafbase + i);
        const double s = res + v;
        res = (i < rem) ? s : res;
    }
    return res;
}

// REG: full 8192-element chunks -- 64 leaves of 128 elements, tree height 6, no
// ragged leaves, no masks; !REG: the last, shorter chunk of a chain (any length).
template <bool UNIT, bool FMA, bool REG, int RNG = BIG_RNG_HBM>
__global__ void __launch_bounds__(256) hmc_gauss_big_traj_kernel(const BigArgs a)
{
    constexpr int GS = 8;                    // a lane's 16 elements, in two halves
    __shared__ double S[4][128];
    __shared__ int dep[128];
    __shared__ double zx[RNG == BIG_RNG_HBM ? 1 : XZIG_C + 1];
    if (RNG != BIG_RNG_HBM) {
        xzig_load_table(zx, threadIdx.x, 256);
        __syncthreads();
    }
    const int H = a.H;
    const int npaths = 1 << H;
    const int lane = threadIdx.x & 63;
    const int j = lane & 7;
    const int group = threadIdx.x >> 3;      // 32 groups of 8 lanes
    const int64_t c = blockIdx.x / a.chunks_here;
    const int chunk = a.chunk0 + (int)(blockIdx.x % a.chunks_here);
    const int64_t cbase = (int64_t)chunk * NPY_BUFSIZE;
    const int n = REG ? NPY_BUFSIZE
                      : ((a.D - cbase < NPY_BUFSIZE) ? (int)(a.D - cbase) : NPY_BUFSIZE);
    const double dt = a.dt_chain ? a.dt_chain[c] : a.timestep;
    const double hdt = 0.5 * dt;
    const double *q0 = a.q0 + c * a.D + cbase;
    const double *p0 = a.p0 + c * a.D + cbase;
    double *qo = a.q_out + c * a.D + cbase;
    // one random stream per lane and (chain, chunk): it serves the lane's leaves in
    // the order the workgroup walks them (lane group g: paths g, g + 32, ...)
    Xo128 gen = {0u, 0u, 0u, 0u};
    if (RNG != BIG_RNG_HBM)
        gen = xo_seed((((uint64_t)(c + a.chain_offset) * (uint64_t)a.nchunks + (uint64_t)chunk) * 32 + (uint64_t)group) * 8
                          + (uint64_t)j, a.rng_seed, a.rng_offset);

    // A lane integrates its leaf in two halves of GS = 8 elements.  The halves of
    // consecutive leaves form one stream of work items through a two-deep
    // register ring (A = first halves, B = second halves): while one half runs
    // its L steps, the loads of the next are in flight.
    struct Item {
        Leaf L;
        bool act, work;
    };
    auto item_of = [&](int base) {
        Item it;
        const int path = base + group;
        it.act = path < npaths;
        if (REG) {
            it.L.off = path * PW_BLOCK; it.L.len = PW_BLOCK; it.L.depth = 6; it.L.canonical = 1;
            it.work = true;
            return it;
        }
        it.L = pairwise_leaf(n, H, it.act ? path : 0);
        // a leaf above depth H is reached by several paths: only the lowest one
        // integrates it, the others copy its sums below
        it.work = it.act && it.L.canonical;
        return it;
    };
    auto load_half = [&](double(&q)[GS], double(&p)[GS], const Item &it, int h) {
#pragma unroll
        for (int i = 0; i < GS; ++i) {
            const int e = 8 * (h * GS + i) + j;
            const bool m = REG || (it.work && (e < it.L.len));
            q[i] = (m && RNG != BIG_RNG_DUMP) ? q0[it.L.off + e] : 0.0;
            if (RNG == BIG_RNG_HBM) p[i] = m ? p0[it.L.off + e] : 0.0;
        }
    };
    LaneSum sq0, sp0, sqL, spL;
    auto run_half = [&](double(&q)[GS], double(&p)[GS], const Item &it, int h, int T) {
        if (RNG != BIG_RNG_HBM) {
            // this half's momentum draw, np.random.normal (hmc.py:146)
            unsigned want = 0;
#pragma unroll
            for (int i = 0; i < GS; ++i)
                if (REG || (it.work && (8 * (h * GS + i) + j < it.L.len))) want |= 1u << i;
            xzig_normals<GS>(p, want, gen, zx);
            if (RNG == BIG_RNG_DUMP) {
                double *po = a.p_dump + c * a.D + cbase;
#pragma unroll
                for (int i = 0; i < GS; ++i)
                    if (want & (1u << i)) po[it.L.off + 8 * (h * GS + i) + j] = p[i];
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < GS; ++i) {                         // hmc.py:143,148
            const double d = UNIT ? q[i] : q[i] - a.x0;
            lane_sum_add<REG>(sq0, d * d, h * GS + i, T);
            lane_sum_add<REG>(sp0, p[i] * p[i], h * GS + i, T);
        }
#pragma unroll
        for (int i = 0; i < GS; ++i)                           // hmc.py:116
            p[i] = kick<FMA>(p[i], hdt, gauss_grad<UNIT>(q[i], a.k, a.x0));
        for (int l = 0; l < a.nsteps - 1; ++l) {               // hmc.py:118-120
#pragma unroll
            for (int i = 0; i < GS; ++i) {
                q[i] = drift<FMA>(q[i], p[i], dt);
                p[i] = kick<FMA>(p[i], dt, gauss_grad<UNIT>(q[i], a.k, a.x0));
            }
        }
#pragma unroll
        for (int i = 0; i < GS; ++i) {                         // hmc.py:122-123
            q[i] = drift<FMA>(q[i], p[i], dt);
            p[i] = kick<FMA>(p[i], hdt, gauss_grad<UNIT>(q[i], a.k, a.x0));
        }
#pragma unroll
        for (int i = 0; i < GS; ++i) {                         // hmc.py:150
            const int t = h * GS + i;
            const double d = UNIT ? q[i] : q[i] - a.x0;
            lane_sum_add<REG>(sqL, d * d, t, T);
            lane_sum_add<REG>(spL, p[i] * p[i], t, T);
            if (REG || (it.work && (8 * t + j < it.L.len))) qo[it.L.off + 8 * t + j] = q[i];
        }
        asm volatile("" : "+v"(sq0.r), "+v"(sp0.r), "+v"(sqL.r), "+v"(spL.r));
    };

    double qa[GS], pa[GS], qb[GS], pb[GS];
    Item cur = item_of(0);
    load_half(qa, pa, cur, 0);
    for (int base = 0; base < npaths; base += 32) {
        const int T = (cur.L.len >= 8) ? (cur.L.len >> 3) : 0;
        const int rem = (cur.L.len >= 8) ? (cur.L.len & 7) : cur.L.len;
        sq0 = {0.0, 0.0}; sp0 = {0.0, 0.0}; sqL = {0.0, 0.0}; spL = {0.0, 0.0};
        load_half(qb, pb, cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        run_half(qa, pa, cur, 0, T);
        __builtin_amdgcn_sched_barrier(0);
        // next leaf's first half (unconditional: past the end it re-reads this
        // leaf and is ignored -- a load under a branch would be waited for at once)
        const Item nxt = item_of(base + 32 < npaths ? base + 32 : base);
        load_half(qa, pa, nxt, 0);
        __builtin_amdgcn_sched_barrier(0);
        run_half(qb, pb, cur, 1, T);
        __builtin_amdgcn_sched_barrier(0);
        const double r0 = leaf_finish<REG>(sq0.r, sq0.tail, T, rem, lane);
        const double r1 = leaf_finish<REG>(sp0.r, sp0.tail, T, rem, lane);
        const double r2 = leaf_finish<REG>(sqL.r, sqL.tail, T, rem, lane);
        const double r3 = leaf_finish<REG>(spL.r, spL.tail, T, rem, lane);
        if (cur.act && j == 0) {
            const int path = base + group;
            dep[path] = cur.L.depth;
            if (cur.work) {
                S[0][path] = r0; S[1][path] = r1; S[2][path] = r2; S[3][path] = r3;
            }
        }
        cur = nxt;
    }
    if (RNG == BIG_RNG_DUMP) return;
    __syncthreads();
    // redundant paths take the sums of the leaf they coincide with
    if (!REG && (int)threadIdx.x < npaths) {
        const int pth = threadIdx.x;
        const int canon = pth & ~((1 << (H - dep[pth])) - 1);
        if (canon != pth) {
#pragma unroll
            for (int v = 0; v < 4; ++v) S[v][pth] = S[v][canon];
        }
    }
    __syncthreads();
    // up the chunk's tree: level l joins the two depth-(H-l) subtrees
    for (int l = 0; l < H; ++l) {
        double v[4] = {0.0, 0.0, 0.0, 0.0};
        const int pth = threadIdx.x;
        if (pth < npaths) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const double mine = S[w][pth];
                v[w] = (dep[pth] >= H - l) ? mine + S[w][pth ^ (1 << l)] : mine;
            }
        }
        __syncthreads();
        if (pth < npaths) {
#pragma unroll
            for (int w = 0; w < 4; ++w) S[w][pth] = v[w];
        }
        __syncthreads();
    }
    if (threadIdx.x < 4)
        a.ws[(c * a.nchunks + chunk) * 4 + threadIdx.x] = S[threadIdx.x][0];
}

struct BigFinishArgs {
    const double *ws;
    const double *u;         // null: the acceptance draw comes from the generator
    double *u_dump;          // write it out (tests)
    uint64_t rng_seed;
    uint64_t rng_offset;
    int64_t chain_offset;
    uint8_t *accepted;
    int64_t *n_accepted;
    double *e_before;
    double *e_after;
    double *dt_chain;
    double k;
    double uprate;
    double downrate;
    int64_t C;
    int32_t nchunks;
    int32_t adapt;
};

__global__ void __launch_bounds__(256) hmc_gauss_big_finish_kernel(const BigFinishArgs a)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= a.C) return;
    double t[4] = {0.0, 0.0, 0.0, 0.0};      // the reduction's identity, then chunk after chunk
    const double *w = a.ws + c * a.nchunks * 4;
    for (int ch = 0; a.ws && ch < a.nchunks; ++ch) {
#pragma unroll
        for (int v = 0; v < 4; ++v) t[v] = t[v] + w[ch * 4 + v];
    }
    const double c_lp = -0.5 * a.k;
    const double Eb = -(c_lp * t[0]) + 0.5 * t[1];            // hmc.py:143,148
    const double Ea = -(c_lp * t[2]) + 0.5 * t[3];            // hmc.py:150
    double x = -(Ea - Eb);                                    // hmc.py:151
    x = (x < -308.0) ? -308.0 : x;
    x = (x > 709.0) ? 709.0 : x;
    double uu;
    if (a.u) {
        uu = a.u[c];
    } else {                                                  // np.random.uniform, hmc.py:151
        Xo128 gen = xo_seed(BIG_U_STREAM + (uint64_t)(c + a.chain_offset), a.rng_seed, a.rng_offset);
        uu = xo_uniform53(gen);
    }
    if (a.u_dump) a.u_dump[c] = uu;
    if (!a.ws) return;                                        // draw dump only
    const bool acc = uu < exp_clipped_range(x);
    a.accepted[c] = acc ? 1 : 0;
    if (a.e_before) a.e_before[c] = Eb;
    if (a.e_after) a.e_after[c] = Ea;
    if (a.n_accepted && acc) a.n_accepted[c] += 1;
    if (a.adapt) {                                            // hmc.py:188-191
        const double dt = a.dt_chain[c];
        a.dt_chain[c] = acc ? dt * a.uprate : dt * a.downrate;
    }
}

// q_out[c, :] = q0[c, :] for rejected chains (hmc.py:164).  A workgroup serves
// one 4096-element segment of one chain: accepted chains cost one flag read per
// workgroup and no traffic.  VEC doubles per access.
constexpr int RESTORE_SEG = 4096;

template <int VEC>
__global__ void __launch_bounds__(256)
restore_rejected_kernel(double *q_out, const double *q0, const uint8_t *accepted, int64_t D,
                        int32_t segs)
{
    const int64_t c = blockIdx.x / segs;
    if (accepted[c]) return;
    const int64_t lo = (int64_t)(blockIdx.x % segs) * RESTORE_SEG;
    const int64_t hi = (lo + RESTORE_SEG < D) ? lo + RESTORE_SEG : D;
    const double *src = q0 + c * D;
    double *dst = q_out + c * D;
    for (int64_t i = lo + (int64_t)threadIdx.x * VEC; i < hi; i += 256 * VEC) {
        if (VEC == 2) {
            *reinterpret_cast<double2 *>(dst + i) = *reinterpret_cast<const double2 *>(src + i);
        } else {
            dst[i] = src[i];
        }
    }
}

}  // namespace binf

using namespace binf;

static int32_t big_chunks(int64_t D) { return (int32_t)((D + NPY_BUFSIZE - 1) / NPY_BUFSIZE); }

extern "C" int64_t binf_hmc_sample_gauss_big_workspace_bytes(int64_t C, int64_t D)
{
    if (C <= 0 || D <= 0) return 0;
    return C * (int64_t)big_chunks(D) * 4 * (int64_t)sizeof(double);
}

template <bool REG, int RNG>
static void big_launch_traj(const BigArgs &a, bool unit, bool fma, dim3 grid, hipStream_t st)
{
    if (RNG == BIG_RNG_DUMP) {
        hmc_gauss_big_traj_kernel<true, false, REG, RNG><<<grid, 256, 0, st>>>(a);
    } else if (unit) {
        if (fma) hmc_gauss_big_traj_kernel<true, true, REG, RNG><<<grid, 256, 0, st>>>(a);
        else     hmc_gauss_big_traj_kernel<true, false, REG, RNG><<<grid, 256, 0, st>>>(a);
    } else {
        if (fma) hmc_gauss_big_traj_kernel<false, true, REG, RNG><<<grid, 256, 0, st>>>(a);
        else     hmc_gauss_big_traj_kernel<false, false, REG, RNG><<<grid, 256, 0, st>>>(a);
    }
}

// The three launches (trajectory per chunk, finish per chain, restore) for one of
// the draw sources: RNG = 0 draws read from p0 / u; 1 generated in the kernels
// under (seed, offset); 2 only written to p_dump / u_dump.
template <int RNG>
static int32_t big_run(const char *what, const double *q0, const double *p0, const double *u,
                       double *q_out, uint8_t *accepted, int64_t *n_accepted, double *e_before,
                       double *e_after, double timestep, double *dt_chain, int64_t C, int64_t D,
                       int32_t nsteps, double k, double x0, int32_t adapt, double uprate,
                       double downrate, int32_t mode, void *workspace, uint64_t seed,
                       uint64_t offset, int64_t chain_offset, double *p_dump, double *u_dump,
                       hipStream_t st)
{
    BigArgs a;
    a.q0 = q0; a.p0 = p0; a.q_out = q_out; a.ws = (double *)workspace; a.dt_chain = dt_chain;
    a.timestep = timestep; a.k = k; a.x0 = x0; a.C = C; a.D = D; a.nchunks = big_chunks(D);
    a.nsteps = nsteps; a.rng_seed = seed; a.rng_offset = offset; a.chain_offset = chain_offset;
    a.p_dump = p_dump;
    const int32_t nfull = (int32_t)(D / NPY_BUFSIZE);
    const int32_t ntail = (D % NPY_BUFSIZE) ? 1 : 0;
    if (C * (int64_t)(nfull > 0 ? nfull : 1) > 0x7fffffffLL)
        return fail(BINF_E_UNSUPPORTED, "%s: too many (chain, chunk) pairs", what);
    const bool unit = (k == 1.0 && x0 == 0.0), fma = (mode == BINF_MODE_FMA);
    hipError_t e;
    if (nfull > 0) {
        a.H = 6; a.chunk0 = 0; a.chunks_here = nfull;
        big_launch_traj<true, RNG>(a, unit, fma, dim3((unsigned)(C * nfull)), st);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "hmc_gauss_big_traj_kernel launch");
    }
    if (ntail) {
        a.H = pairwise_tree_height(D % NPY_BUFSIZE);
        if (a.H > 7) return fail(BINF_E_UNSUPPORTED, "%s: pairwise tree height %d", what, a.H);
        a.chunk0 = nfull; a.chunks_here = 1;
        big_launch_traj<false, RNG>(a, unit, fma, dim3((unsigned)C), st);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "hmc_gauss_big_traj_kernel (tail chunk) launch");
    }
    BigFinishArgs f;
    f.ws = (RNG == BIG_RNG_DUMP) ? nullptr : (const double *)workspace;
    f.u = (RNG == BIG_RNG_HBM) ? u : nullptr; f.u_dump = u_dump; f.rng_seed = seed;
    f.rng_offset = offset; f.chain_offset = chain_offset; f.accepted = accepted; f.n_accepted = n_accepted;
    f.e_before = e_before; f.e_after = e_after; f.dt_chain = dt_chain; f.k = k;
    f.uprate = uprate; f.downrate = downrate; f.C = C; f.nchunks = a.nchunks;
    f.adapt = adapt ? 1 : 0;
    hmc_gauss_big_finish_kernel<<<dim3((unsigned)((C + 255) / 256)), 256, 0, st>>>(f);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "hmc_gauss_big_finish_kernel launch");
    if (RNG == BIG_RNG_DUMP) return 0;
    const bool vec2 = (D % 2 == 0) && ((((uintptr_t)q_out | (uintptr_t)q0) & 15) == 0);
    const int32_t segs = (int32_t)((D + RESTORE_SEG - 1) / RESTORE_SEG);
    if (C * segs > 0x7fffffffLL) return fail(BINF_E_UNSUPPORTED, "%s: too many segments", what);
    const dim3 rgrid((unsigned)(C * segs));
    if (vec2) restore_rejected_kernel<2><<<rgrid, 256, 0, st>>>(q_out, q0, accepted, D, segs);
    else      restore_rejected_kernel<1><<<rgrid, 256, 0, st>>>(q_out, q0, accepted, D, segs);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "restore_rejected_kernel launch");
    return 0;
}

static int32_t big_check(const char *what, const double *q0, const double *p0, double *q_out,
                         const uint8_t *accepted, double *dt_chain, int64_t C, int64_t D,
                         int32_t nsteps, int32_t adapt, int32_t mode, const void *workspace,
                         int64_t workspace_bytes)
{
    if (C < 0 || D < 1 || nsteps < 1) return fail(BINF_E_ARG, "%s: need C>=0, D>=1, nsteps>=1", what);
    if (mode != BINF_MODE_EXACT && mode != BINF_MODE_FMA)
        return fail(BINF_E_ARG, "%s: unknown mode %d", what, mode);
    if (C == 0) return 0;
    if (!q0 || !q_out || !accepted) return fail(BINF_E_ARG, "%s: null buffer", what);
    if (adapt && !dt_chain) return fail(BINF_E_ARG, "%s: adaption needs dt_chain", what);
    if (C > 0x7fffffffffffffffLL / D) return fail(BINF_E_ARG, "%s: C*D overflows", what);
    const int64_t bytes = C * D * (int64_t)sizeof(double);
    const char *qo = (const char *)q_out, *qi = (const char *)q0, *pi = (const char *)p0;
    if ((qo < qi + bytes && qi < qo + bytes) || (pi && qo < pi + bytes && pi < qo + bytes))
        return fail(BINF_E_ALIAS, "%s: q_out must not overlap q0 or p0 (rejected chains are "
                    "restored from q0)", what);
    const int64_t need = binf_hmc_sample_gauss_big_workspace_bytes(C, D);
    if (!workspace || workspace_bytes < need)
        return fail(BINF_E_ARG, "%s: needs %lld bytes of workspace, got %lld", what,
                    (long long)need, (long long)workspace_bytes);
    return 0;
}

extern "C" int32_t binf_hmc_sample_gauss_big_f64(
    const double *q0, const double *p0, const double *u, double *q_out, uint8_t *accepted,
    int64_t *n_accepted, double *e_before, double *e_after, double timestep, double *dt_chain,
    int64_t C, int64_t D, int32_t nsteps, double k, double x0, int32_t adapt, double uprate,
    double downrate, int32_t mode, void *workspace, int64_t workspace_bytes, void *stream)
{
    const char *what = "hmc_sample_gauss_big";
    if (C > 0 && D >= 1 && nsteps >= 1 && (!p0 || !u)) return fail(BINF_E_ARG, "%s: null buffer", what);
    if (int32_t rc = big_check(what, q0, p0, q_out, accepted, dt_chain, C, D, nsteps, adapt, mode,
                               workspace, workspace_bytes)) return rc;
    if (C == 0) return 0;
    return big_run<BIG_RNG_HBM>(what, q0, p0, u, q_out, accepted, n_accepted, e_before, e_after,
                                timestep, dt_chain, C, D, nsteps, k, x0, adapt, uprate, downrate,
                                mode, workspace, 0, 0, 0, nullptr, nullptr, (hipStream_t)stream);
}

// The same transition with its draws generated inside the kernels (momentum: one
// xoshiro128++ stream per lane and (chain, chunk); acceptance draw: one per chain
// in the finish kernel), keyed by (seed, offset): no momentum buffer.
extern "C" int32_t binf_hmc_sample_gauss_big_rng_f64(
    const double *q0, double *q_out, uint8_t *accepted, int64_t *n_accepted, double *e_before,
    double *e_after, double timestep, double *dt_chain, int64_t C, int64_t D, int32_t nsteps,
    double k, double x0, int32_t adapt, double uprate, double downrate, int32_t mode,
    uint64_t seed, uint64_t offset, int64_t chain_offset, void *workspace,
    int64_t workspace_bytes, void *stream)
{
    const char *what = "hmc_sample_gauss_big_rng";
    if (chain_offset < 0) return fail(BINF_E_ARG, "%s: chain_offset < 0", what);
    if (int32_t rc = big_check(what, q0, nullptr, q_out, accepted, dt_chain, C, D, nsteps, adapt,
                               mode, workspace, workspace_bytes)) return rc;
    if (C == 0) return 0;
    return big_run<BIG_RNG_FUSED>(what, q0, nullptr, nullptr, q_out, accepted, n_accepted, e_before,
                                  e_after, timestep, dt_chain, C, D, nsteps, k, x0, adapt, uprate,
                                  downrate, mode, workspace, seed, offset, chain_offset, nullptr,
                                  nullptr,
                                  (hipStream_t)stream);
}

// The draws binf_hmc_sample_gauss_big_rng_f64 consumes for (seed, offset, C, D),
// written out: p0_out [C*D], u_out [C].
extern "C" int32_t binf_hmc_gauss_big_rng_draws_f64(double *p0_out, double *u_out, int64_t C,
                                                    int64_t D, uint64_t seed, uint64_t offset,
                                                    int64_t chain_offset, void *stream)
{
    const char *what = "hmc_gauss_big_rng_draws";
    if (C < 0 || D < 1 || chain_offset < 0)
        return fail(BINF_E_ARG, "%s: need C>=0, D>=1, chain_offset>=0", what);
    if (C == 0) return 0;
    if (!p0_out || !u_out) return fail(BINF_E_ARG, "%s: null buffer", what);
    if (C > 0x7fffffffffffffffLL / D) return fail(BINF_E_ARG, "%s: C*D overflows", what);
    return big_run<BIG_RNG_DUMP>(what, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                 nullptr, 0.0, nullptr, C, D, 1, 1.0, 0.0, 0, 1.0, 1.0,
                                 BINF_MODE_EXACT, nullptr, seed, offset, chain_offset, p0_out, u_out,
                                 (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// n transitions from one call: the `for i in range(n): sampler.sample()` loop of
// example_script.py:33-34 for long chains.  The trajectory kernel moves its 24 D
// bytes per chain as fast per byte as the persistent kernel of hmc_gauss.hip moves
// its 16 D (measured: 0.22 vs 0.245 ps/B), so there is no persistent variant to
// gain from for chains that do not fit a workgroup's registers; what a loop of
// single calls DOES waste is the copy of every recorded state (torch.stack: +16 D)
// and the host between the launches.  Here transition s writes its proposal
// straight into the record slot (or into one of two scratch states), rejected
// chains are restored from the state the transition read, and the next transition
// reads what this one wrote: 3 launches and 24 D bytes per transition, recorded
// or not.  Bit-identical to n single calls (the same kernels).
// ---------------------------------------------------------------------------
extern "C" int64_t binf_hmc_sample_n_gauss_big_workspace_bytes(int64_t C, int64_t D)
{
    if (C <= 0 || D <= 0) return 0;
    return binf_hmc_sample_gauss_big_workspace_bytes(C, D) + C * D * (int64_t)sizeof(double);
}

template <int RNG>
static int32_t big_run_n(const char *what, const double *q0, const double *p0, const double *u,
                         double *q_out, double *samples, uint8_t *accepted, int64_t *n_accepted,
                         double *e_before, double *e_after, double timestep, double *dt_chain,
                         int64_t C, int64_t D, int32_t nsteps, int32_t n, int32_t thin, double k,
                         double x0, int32_t n_adapt, double uprate, double downrate, int32_t mode,
                         void *workspace, int64_t workspace_bytes, uint64_t seed, uint64_t offset,
                         int64_t chain_offset, hipStream_t st)
{
    if (n < 1 || thin < 1 || n_adapt < 0) return fail(BINF_E_ARG, "%s: need n>=1, thin>=1, n_adapt>=0", what);
    if (int32_t rc = big_check(what, q0, p0, q_out, accepted, dt_chain, C, D, nsteps, n_adapt > 0,
                               mode, workspace, workspace_bytes)) return rc;
    if (C == 0) return 0;
    if (workspace_bytes < binf_hmc_sample_n_gauss_big_workspace_bytes(C, D))
        return fail(BINF_E_ARG, "%s: needs %lld bytes of workspace, got %lld", what,
                    (long long)binf_hmc_sample_n_gauss_big_workspace_bytes(C, D), (long long)workspace_bytes);
    const int64_t CD = C * D;
    // Every proposal is written straight to where it is kept (a `samples` slot, q_out or the
    // scratch state) and a rejected chain is restored from the buffer its transition read:
    // the record buffer must not touch q0, q_out, the [n x C x D] momenta or the workspace,
    // and no momentum block (not only the first, which big_check looks at) may touch q_out.
    const int64_t nrec = samples ? n / thin : 0;
    if (nrec > 0 && (overlap_f64(samples, nrec * CD, q0, CD) || overlap_f64(samples, nrec * CD, q_out, CD) ||
                     (p0 && overlap_f64(samples, nrec * CD, p0, (int64_t)n * CD)) ||
                     overlap_f64(samples, nrec * CD, workspace, (workspace_bytes + 7) / 8)))
        return fail(BINF_E_ALIAS, "%s: samples overlaps q0, q_out, p0 or the workspace", what);
    if (p0 && overlap_f64(q_out, CD, p0, (int64_t)n * CD))
        return fail(BINF_E_ALIAS, "%s: q_out overlaps the [n x C x D] momenta", what);
    if (overlap_f64(q0, CD, workspace, (workspace_bytes + 7) / 8) ||
        overlap_f64(q_out, CD, workspace, (workspace_bytes + 7) / 8))
        return fail(BINF_E_ALIAS, "%s: q0 / q_out overlap the workspace", what);
    double *scratch = (double *)((char *)workspace + binf_hmc_sample_gauss_big_workspace_bytes(C, D));
    // unrecorded transitions alternate between q_out and the scratch state so that
    // the LAST unrecorded one of a run lands in q_out
    int n_unrec = 0;
    for (int s = 0; s < n; ++s)
        if (!(samples && (s + 1) % thin == 0)) ++n_unrec;
    int unrec_seen = 0;
    const double *in = q0;
    double *out = nullptr;
    for (int s = 0; s < n; ++s) {
        const bool rec = samples && (s + 1) % thin == 0;
        if (rec) {
            out = samples + (int64_t)((s + 1) / thin - 1) * CD;
        } else {
            // the k-th unrecorded transition from the end writes q_out iff k is odd
            out = ((n_unrec - unrec_seen) & 1) ? q_out : scratch;
            ++unrec_seen;
        }
        const int32_t rc = big_run<RNG>(
            what, in, p0 ? p0 + (int64_t)s * CD : nullptr, u ? u + (int64_t)s * C : nullptr, out,
            accepted + (int64_t)s * C, n_accepted, e_before ? e_before + (int64_t)s * C : nullptr,
            e_after ? e_after + (int64_t)s * C : nullptr, timestep, dt_chain, C, D, nsteps, k, x0,
            s < n_adapt ? 1 : 0, uprate, downrate, mode, workspace, seed, offset + (uint64_t)s,
            chain_offset, nullptr, nullptr, st);
        if (rc) return rc;
        in = out;
    }
    if (out != q_out) {
        const hipError_t e = hipMemcpyAsync(q_out, out, (size_t)CD * sizeof(double),
                                            hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return hip_fail(e, "final state copy");
    }
    return 0;
}

extern "C" int32_t binf_hmc_sample_n_gauss_big_f64(
    const double *q0, const double *p0, const double *u, double *q_out, double *samples,
    uint8_t *accepted, int64_t *n_accepted, double *e_before, double *e_after, double timestep,
    double *dt_chain, int64_t C, int64_t D, int32_t nsteps, int32_t n, int32_t thin, double k,
    double x0, int32_t n_adapt, double uprate, double downrate, int32_t mode, void *workspace,
    int64_t workspace_bytes, void *stream)
{
    const char *what = "hmc_sample_n_gauss_big";
    if (C > 0 && D >= 1 && nsteps >= 1 && (!p0 || !u)) return fail(BINF_E_ARG, "%s: null buffer", what);
    return big_run_n<BIG_RNG_HBM>(what, q0, p0, u, q_out, samples, accepted, n_accepted, e_before,
                                  e_after, timestep, dt_chain, C, D, nsteps, n, thin, k, x0, n_adapt,
                                  uprate, downrate, mode, workspace, workspace_bytes, 0, 0, 0,
                                  (hipStream_t)stream);
}

extern "C" int32_t binf_hmc_sample_n_gauss_big_rng_f64(
    const double *q0, double *q_out, double *samples, uint8_t *accepted, int64_t *n_accepted,
    double *e_before, double *e_after, double timestep, double *dt_chain, int64_t C, int64_t D,
    int32_t nsteps, int32_t n, int32_t thin, double k, double x0, int32_t n_adapt, double uprate,
    double downrate, int32_t mode, uint64_t seed, uint64_t offset, int64_t chain_offset,
    void *workspace, int64_t workspace_bytes, void *stream)
{
    const char *what = "hmc_sample_n_gauss_big_rng";
    if (chain_offset < 0) return fail(BINF_E_ARG, "%s: chain_offset < 0", what);
    return big_run_n<BIG_RNG_FUSED>(what, q0, nullptr, nullptr, q_out, samples, accepted, n_accepted,
                                    e_before, e_after, timestep, dt_chain, C, D, nsteps, n, thin, k,
                                    x0, n_adapt, uprate, downrate, mode, workspace, workspace_bytes,
                                    seed, offset, chain_offset, (hipStream_t)stream);
}

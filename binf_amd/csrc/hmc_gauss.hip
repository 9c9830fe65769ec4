// Fused HMC on an isotropic Gaussian (the reference's TestHO,
// binf/pdf/__init__.py:181-191): ONE launch = n consecutive HMCSampler.sample()
// transitions of every chain (n = 1: binf_hmc_sample_gauss_f64; n > 1: the
// `for i in range(n): sampler.sample()` loop of example_script.py:33-34).
// Replaces binf/samplers/hmc.py:92-164,183-191.  gfx950 (MI355X), wave64.
//
// Mapping.  A chain's D coordinates are owned by G = 8 * 2^H lanes of ONE wave
// (H = height of numpy's pairwise-sum tree for length D), so 64/G chains share
// a wave.  Lane (leaf path g, accumulator j) owns elements off_g + 8t + j,
// t = 0..TMAX-1 -- exactly the elements numpy's j-th strided accumulator of
// that leaf adds up, in order -- so the energy reductions are an in-lane
// running sum, xor-shuffles 1,2,4 inside the leaf (numpy's
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))), the leaf's tail elements, then
// xor-shuffles 8,16,32 up the leaf tree: bit-identical to np.sum, no LDS.
// HBM accesses are 8 bytes per lane in 64-byte segments (measured at the same
// 6.3 TB/s as 16-byte coalesced streaming).
//
// Where the state lives between transitions:
//   * q stays in VGPRs across transitions (no q0 re-read, no q_out write unless
//     the draw is recorded);
//   * the state before the transition is stashed in LDS (8 KiB per wave) and
//     read back only on rejection;
//   * V(q) of the current state is carried over instead of being re-reduced
//     (re-reducing the same bits gives the same bits);
//   * the Gaussian is separable, so each group of GS (8) elements per lane runs
//     its whole trajectory on its own; the momentum streams in one group at a
//     time, the next group's draw (or the next transition's first group)
//     being fetched while the current group integrates.
#include "gauss_common.hpp"

namespace binf {

// LW = log2(waves per chain).  LW = 0: a chain is G = 8 << H <= 64 lanes of one
// wave (several chains per wave when G < 64).  LW > 0 (D > 1024): a chain spans
// 2 / 4 / 8 whole waves of the workgroup; the leaf-tree levels above a wave are
// joined through LDS (chain_sum_finish).
template <int TMAX, bool REGULAR, bool UNIT, bool FMA, int LW>
__global__ void __launch_bounds__(LW == 3 ? 512 : 256)
hmc_gauss_persist_kernel(const GaussNArgs a)
{
    constexpr int WPB = (LW == 3) ? 8 : 4;           // waves per workgroup
    constexpr int WPC = 1 << LW;                     // waves per chain
    __shared__ double xch[WPB];
    constexpr int GS = (TMAX % 8 == 0) ? 8 : ((TMAX % 4 == 0) ? 4 : TMAX);   // measured: 8 beats 4 and 16
    constexpr int NG = TMAX / GS;
    __shared__ double stash[WPB][TMAX][64];

    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * WPB + wib;
    const int H = a.H;
    const int lg = (LW > 0) ? 6 : 3 + H;             // log2(lanes of a chain in this wave)
    const int slot = lane & ((1 << lg) - 1);
    const int j = slot & 7;
    const int wchain = wib & (WPC - 1);              // wave index inside the chain
    const int grp = (LW > 0) ? ((wchain << 3) | (lane >> 3)) : (slot >> 3);
    const bool writer = (LW > 0) ? (wchain == 0 && lane == 0) : (slot == 0);

    int off, n, leafdepth, canonical;
    if (REGULAR) {
        n = 8 * TMAX;
        off = grp * n;
        leafdepth = H;
        canonical = 1;
    } else {
        const Leaf L = pairwise_leaf(a.D, H, grp);
        off = L.off;
        n = L.len;
        leafdepth = L.depth;
        canonical = L.canonical;
    }
    const int T = (n >= 8) ? (n >> 3) : 0;
    const int rem = (n >= 8) ? (n & 7) : n;

    const int64_t raw = (LW > 0) ? (int64_t)blockIdx.x * (WPB / WPC) + (wib >> LW)
                                 : (wave << (6 - lg)) + (lane >> lg);
    const bool cvalid = raw < a.C;
    const int64_t chain = cvalid ? raw : a.C - 1;
    const int64_t CD = a.C * (int64_t)a.D;
    const int64_t base = chain * (int64_t)a.D + off + j;

    double dt = a.dt_chain ? a.dt_chain[chain] : a.timestep;
    double uu = a.u[chain];
    if (a.stagger > 0) {
        // De-phase the waves that share a SIMD: a launch puts every wave in the
        // same phase (all load, then all integrate, then all store), so the
        // memory pipe idles while the FP64 pipe works and vice versa.  Wave slot
        // s of its SIMD (HW_ID.WAVE_ID) starts s * stagger * 64 cycles late.
        const int slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 3;
        for (int i = 0; i < slot * a.stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_sched_barrier(0);

    // q lives in registers for the whole launch; the momentum is needed one
    // element group at a time, so it streams through a 2-deep register ring
    // (pa / pb): while group g runs its trajectory, group g+1's draw (or group
    // 0 of the next transition) is in flight.
    double q[TMAX], pa[GS], pb[GS];
    // issue order = arrival order: the first group's state and momentum first,
    // so its trajectory can start while the rest of the state is in flight
#pragma unroll
    for (int t = 0; t < GS; ++t) {
        const bool m = REGULAR || (8 * t + j < n);
        q[t] = m ? a.q0[base + 8 * t] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < GS; ++i) {
        const bool m = REGULAR || (8 * i + j < n);
        pa[i] = m ? a.p0[base + 8 * i] : 0.0;
    }
#pragma unroll
    for (int t = GS; t < TMAX; ++t) {
        const bool m = REGULAR || (8 * t + j < n);
        q[t] = m ? a.q0[base + 8 * t] : 0.0;
    }

    const double c_lp = -0.5 * a.k;
    // np.sum((q - x0)**2) of the CURRENT state, carried across transitions
    // (for the start state it is summed group by group inside the first
    // transition, so that the first trajectories need not wait for all of q0)
    LaneSum s0 = {0.0, 0.0};
    double Sq_state = 0.0;
    int64_t nacc = 0;

    for (int s = 0; s < a.n; ++s) {
        const double hdt = 0.5 * dt;
        // state before the transition -> LDS (read back only on rejection)
#pragma unroll
        for (int t = 0; t < TMAX; ++t) stash[wib][t][lane] = q[t];

        // Prefetches are issued UNCONDITIONALLY (on the last transition they
        // re-read this transition's data and are ignored): a load under a
        // branch makes the compiler's vmcnt bookkeeping assume it may not have
        // been issued, and the next counted wait then also waits for it.
        const bool more = s + 1 < a.n;
        const double *pc = a.p0 + (int64_t)s * CD + base;
        const double *pn = more ? pc + CD : pc;
        const double un = a.u[(int64_t)(more ? s + 1 : s) * a.C + chain];

        LaneSum spb = {0.0, 0.0}, sqa = {0.0, 0.0}, spa = {0.0, 0.0};
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            double(&cur)[GS] = (g & 1) ? pb : pa;
            double(&nxt)[GS] = (g & 1) ? pa : pb;
            // fetch the next group's draw (next transition's group 0 at the end)
            if (g + 1 < NG) {
#pragma unroll
                for (int i = 0; i < GS; ++i) {
                    const int t = (g + 1) * GS + i;
                    const bool m = REGULAR || (8 * t + j < n);
                    nxt[i] = m ? pc[8 * t] : 0.0;
                }
            } else {
#pragma unroll
                for (int i = 0; i < GS; ++i) {
                    const bool m = REGULAR || (8 * i + j < n);
                    nxt[i] = m ? pn[8 * i] : 0.0;
                }
            }
            // pin this group's values to this point: without it the compiler
            // forms the p*p / q*q products of every group early and keeps
            // them alive until the group's turn
#pragma unroll
            for (int i = 0; i < GS; ++i)
                asm volatile("" : "+v"(cur[i]), "+v"(q[g * GS + i]));
            if (s == 0) {
#pragma unroll
                for (int i = 0; i < GS; ++i) {
                    const int t = g * GS + i;
                    const double d = UNIT ? q[t] : q[t] - a.x0;
                    lane_sum_add<REGULAR>(s0, d * d, t, T);
                }
            }
#pragma unroll
            for (int i = 0; i < GS; ++i)                      // hmc.py:148
                lane_sum_add<REGULAR>(spb, cur[i] * cur[i], g * GS + i, T);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < GS; ++i) {                    // hmc.py:116
                const int t = g * GS + i;
                cur[i] = kick<FMA>(cur[i], hdt, gauss_grad<UNIT>(q[t], a.k, a.x0));
            }
            for (int l = 0; l < a.nsteps - 1; ++l) {          // hmc.py:118-120
#pragma unroll
                for (int i = 0; i < GS; ++i) {
                    const int t = g * GS + i;
                    q[t] = drift<FMA>(q[t], cur[i], dt);
                    cur[i] = kick<FMA>(cur[i], dt, gauss_grad<UNIT>(q[t], a.k, a.x0));
                }
            }
#pragma unroll
            for (int i = 0; i < GS; ++i) {                    // hmc.py:122-123
                const int t = g * GS + i;
                q[t] = drift<FMA>(q[t], cur[i], dt);
                cur[i] = kick<FMA>(cur[i], hdt, gauss_grad<UNIT>(q[t], a.k, a.x0));
            }
#pragma unroll
            for (int i = 0; i < GS; ++i) {                    // hmc.py:150
                const int t = g * GS + i;
                const double d = UNIT ? q[t] : q[t] - a.x0;
                lane_sum_add<REGULAR>(sqa, d * d, t, T);
                lane_sum_add<REGULAR>(spa, cur[i] * cur[i], t, T);
            }
            // ... and pin the running sums here: otherwise the group's last
            // half kick and its squares are sunk below the NEXT group's step
            // loop and its momenta stay live through it
            asm volatile("" : "+v"(sqa.r), "+v"(spa.r), "+v"(spb.r));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (NG & 1) {
            // odd group count: the next transition's group 0 landed in pb
#pragma unroll
            for (int i = 0; i < GS; ++i) pa[i] = pb[i];
        }
        if (s == 0)
            Sq_state = chain_sum_finish<REGULAR, LW>(s0, T, rem, lane, H, leafdepth, xch, wib);
        const double Spb = chain_sum_finish<REGULAR, LW>(spb, T, rem, lane, H, leafdepth, xch, wib);
        const double Sqa = chain_sum_finish<REGULAR, LW>(sqa, T, rem, lane, H, leafdepth, xch, wib);
        const double Spa = chain_sum_finish<REGULAR, LW>(spa, T, rem, lane, H, leafdepth, xch, wib);
        const double Eb = -(c_lp * Sq_state) + 0.5 * Spb;
        const double Ea = -(c_lp * Sqa) + 0.5 * Spa;

        double x = -(Ea - Eb);                                // hmc.py:151
        x = (x < -308.0) ? -308.0 : x;
        x = (x > 709.0) ? 709.0 : x;
        const bool acc = uu < exp_clipped_range(x);

        if (s < a.n_adapt)                                    // hmc.py:188-191
            dt = acc ? dt * a.uprate : dt * a.downrate;
        if (cvalid && writer) {
            const int64_t o = (int64_t)s * a.C + chain;
            if (a.accepted) a.accepted[o] = acc ? 1 : 0;
            if (a.e_before) a.e_before[o] = Eb;
            if (a.e_after) a.e_after[o] = Ea;
        }
        if (acc) {
            Sq_state = Sqa;
            nacc += 1;
        } else {
#pragma unroll
            for (int t = 0; t < TMAX; ++t) q[t] = stash[wib][t][lane];
        }
        if (a.samples && (s + 1) % a.thin == 0 && cvalid && canonical) {
            double *go = a.samples + (int64_t)((s + 1) / a.thin - 1) * CD + base;
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (REGULAR || (8 * t + j < n)) go[8 * t] = q[t];
        }
        uu = un;
    }

    if (cvalid && writer) {
        if (a.n_accepted) a.n_accepted[chain] += nacc;
        if (a.n_adapt > 0 && a.dt_chain) a.dt_chain[chain] = dt;
    }
    if (cvalid && canonical) {
        double *go = a.q_out + base;
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (REGULAR || (8 * t + j < n)) go[8 * t] = q[t];
    }
}

template <int TMAX, bool REGULAR, int LW>
static hipError_t launch_n_trl(const GaussNArgs &a, bool unit, bool fma, dim3 grid,
                               hipStream_t st)
{
    constexpr int BS = (LW == 3) ? 512 : 256;
    if (unit) {
        if (fma) hmc_gauss_persist_kernel<TMAX, REGULAR, true, true, LW><<<grid, BS, 0, st>>>(a);
        else     hmc_gauss_persist_kernel<TMAX, REGULAR, true, false, LW><<<grid, BS, 0, st>>>(a);
    } else {
        if (fma) hmc_gauss_persist_kernel<TMAX, REGULAR, false, true, LW><<<grid, BS, 0, st>>>(a);
        else     hmc_gauss_persist_kernel<TMAX, REGULAR, false, false, LW><<<grid, BS, 0, st>>>(a);
    }
    return hipGetLastError();
}

template <int TMAX>
static hipError_t launch_n_t(const GaussNArgs &a, bool regular, bool unit, bool fma,
                             dim3 grid, hipStream_t st)
{
    return regular ? launch_n_trl<TMAX, true, 0>(a, unit, fma, grid, st)
                   : launch_n_trl<TMAX, false, 0>(a, unit, fma, grid, st);
}

// chains spanning 2^LW waves: leaves of any length <= 128, so TMAX = 16
template <int LW>
static hipError_t launch_n_wide(const GaussNArgs &a, bool regular, bool unit, bool fma,
                                dim3 grid, hipStream_t st)
{
    return regular ? launch_n_trl<16, true, LW>(a, unit, fma, grid, st)
                   : launch_n_trl<16, false, LW>(a, unit, fma, grid, st);
}

// start stagger of the waves of a SIMD (units of ~64 cycles per wave slot)
int gauss_stagger(int n)
{
    static int forced = -2;
    if (forced == -2) {
        const char *e = getenv("BINF_GAUSS_STAGGER");        // development aid
        forced = e ? atoi(e) : -1;
    }
    if (forced >= 0) return forced;
    return 0;
}

}  // namespace binf

using namespace binf;

extern "C" int32_t binf_hmc_sample_n_gauss_f64(
    const double *q0, const double *p0, const double *u, double *q_out,
    double *samples, uint8_t *accepted, int64_t *n_accepted, double *e_before,
    double *e_after, double timestep, double *dt_chain, int64_t C, int64_t D,
    int32_t nsteps, int32_t n, int32_t thin, double k, double x0,
    int32_t n_adapt, double uprate, double downrate, int32_t mode, void *stream)
{
    if (C < 0 || D < 1 || nsteps < 1 || n < 1 || thin < 1 || n_adapt < 0)
        return fail(BINF_E_ARG, "hmc_sample_gauss: need C>=0, D>=1, nsteps>=1, n>=1, thin>=1, n_adapt>=0");
    if (mode != BINF_MODE_EXACT && mode != BINF_MODE_FMA)
        return fail(BINF_E_ARG, "hmc_sample_gauss: unknown mode %d", mode);
    if (C == 0) return 0;
    if (!q0 || !p0 || !u || !q_out)
        return fail(BINF_E_ARG, "hmc_sample_gauss: null buffer");
    if (n_adapt > 0 && !dt_chain)
        return fail(BINF_E_ARG, "hmc_sample_gauss: adaption needs dt_chain");
    if (D > 8192)
        return fail(BINF_E_UNSUPPORTED, "hmc_sample_gauss: D=%lld > 8192 not covered by the fused kernel", (long long)D);
    const int32_t H = pairwise_tree_height(D);
    if (H > 6)
        return fail(BINF_E_UNSUPPORTED, "hmc_sample_gauss: pairwise tree height %d > 6 for D=%lld", H, (long long)D);
    const int LW = H > 3 ? H - 3 : 0;                // log2(waves per chain)
    const int64_t bytes = C * D * (int64_t)sizeof(double);
    const char *qo = (const char *)q_out, *qi = (const char *)q0, *pi = (const char *)p0;
    if ((qo != qi && qo < qi + bytes && qi < qo + bytes) ||
        (qo < pi + bytes * n && pi < qo + bytes))
        return fail(BINF_E_ALIAS, "hmc_sample_gauss: q_out overlaps q0/p0 (only q_out == q0 is allowed)");

    int tneed = 1;
    bool regular = true;
    int32_t len0 = -1;
    for (int g = 0; g < (1 << H); ++g) {
        Leaf L = pairwise_leaf((int32_t)D, H, g);
        int tn = (L.len + 7) / 8;
        if (tn > tneed) tneed = tn;
        if (len0 < 0) len0 = L.len;
        if (L.len != len0 || L.depth != H || (L.len & 7)) regular = false;
    }
    GaussNArgs a;
    a.q0 = q0; a.p0 = p0; a.u = u; a.q_out = q_out; a.samples = samples;
    a.accepted = accepted; a.n_accepted = n_accepted; a.e_before = e_before;
    a.e_after = e_after; a.dt_chain = dt_chain; a.timestep = timestep; a.k = k;
    a.x0 = x0; a.uprate = uprate; a.downrate = downrate; a.C = C;
    a.D = (int32_t)D; a.nsteps = nsteps; a.H = H; a.n = n; a.thin = thin;
    a.n_adapt = n_adapt < n ? n_adapt : n;
    a.stagger = gauss_stagger(n);

    int64_t blocks;
    if (LW == 0) {
        const int64_t chains_per_wave = 64 >> (3 + H);
        const int64_t waves = (C + chains_per_wave - 1) / chains_per_wave;
        blocks = (waves + 3) / 4;
    } else {
        const int64_t chains_per_block = (LW == 3 ? 8 : 4) >> LW;
        blocks = (C + chains_per_block - 1) / chains_per_block;
    }
    if (blocks > 0x7fffffffLL)
        return fail(BINF_E_UNSUPPORTED, "hmc_sample_gauss: too many chains");
    dim3 grid((unsigned)blocks);
    hipStream_t st = (hipStream_t)stream;
    const bool unit = (k == 1.0 && x0 == 0.0);
    const bool fma = (mode == BINF_MODE_FMA);
    hipError_t e;
    const int split = (LW == 0) ? gauss_split_factor(C, H, regular, tneed) : 1;
    if (split > 1)        e = launch_gauss_split(a, tneed, split, unit, fma, st);
    else if (LW == 1)     e = launch_n_wide<1>(a, regular && tneed == 16, unit, fma, grid, st);
    else if (LW == 2)     e = launch_n_wide<2>(a, regular && tneed == 16, unit, fma, grid, st);
    else if (LW == 3)     e = launch_n_wide<3>(a, regular && tneed == 16, unit, fma, grid, st);
    else if (tneed <= 1)  e = launch_n_t<1>(a, regular && tneed == 1, unit, fma, grid, st);
    else if (tneed <= 2)  e = launch_n_t<2>(a, regular && tneed == 2, unit, fma, grid, st);
    else if (tneed <= 4)  e = launch_n_t<4>(a, regular && tneed == 4, unit, fma, grid, st);
    else if (tneed <= 8)  e = launch_n_t<8>(a, regular && tneed == 8, unit, fma, grid, st);
    else if (tneed <= 12) e = launch_n_t<12>(a, regular && tneed == 12, unit, fma, grid, st);
    else                  e = launch_n_t<16>(a, regular && tneed == 16, unit, fma, grid, st);
    if (e != hipSuccess) return hip_fail(e, "hmc_gauss_persist_kernel launch");
    return 0;
}

// One transition per launch: HMCSampler.sample() for every chain.
extern "C" int32_t binf_hmc_sample_gauss_f64(
    const double *q0, const double *p0, const double *u, double *q_out,
    uint8_t *accepted, int64_t *n_accepted, double *e_before, double *e_after,
    double timestep, double *dt_chain, int64_t C, int64_t D, int32_t nsteps,
    double k, double x0, int32_t adapt, double uprate, double downrate,
    int32_t mode, void *stream)
{
    if (C > 0 && D >= 1 && nsteps >= 1 && q0 && p0 && u && q_out && !accepted)
        return fail(BINF_E_ARG, "hmc_sample_gauss: null buffer");
    return binf_hmc_sample_n_gauss_f64(q0, p0, u, q_out, nullptr, accepted,
                                       n_accepted, e_before, e_after, timestep,
                                       dt_chain, C, D, nsteps, 1, 1, k, x0,
                                       adapt ? 1 : 0, uprate, downrate, mode,
                                       stream);
}

// hmc_gauss_persist_kernel: the fused Gaussian HMC kernel template, shared by
// hmc_gauss.hip (draws read from HBM) and hmc_gauss_rng.hip (draws generated in
// the kernel).  See hmc_gauss.hip for the mapping and the reference lines.
#pragma once
#include "gauss_common.hpp"
#include "xoshiro.hpp"

namespace binf {

// RNG = 0: the momentum / uniform draws are read from HBM (a.p0, a.u);
// RNG = 1: they are generated in the kernel (xoshiro.hpp), nothing is read;
// RNG = 2: the same draws are only WRITTEN OUT (a.p_dump, a.u_dump) and no
//          trajectory is integrated -- what makes the fused generator testable:
//          sample_n with RNG = 1 must equal sample_n fed with this dump, bit for bit.
enum { GAUSS_RNG_HBM = 0, GAUSS_RNG_FUSED = 1, GAUSS_RNG_DUMP = 2 };

// LW = log2(waves per chain).  LW = 0: a chain is G = 8 << H <= 64 lanes of one
// wave (several chains per wave when G < 64).  LW > 0 (D > 1024): a chain spans
// 2 / 4 / 8 whole waves of the workgroup; the leaf-tree levels above a wave are
// joined through LDS (chain_sum_finish).
// Waves per workgroup: 4 (8 for chains of 8 waves).  With the generator in the
// kernel: 8, so that two workgroups per CU share the 160 KiB of LDS as 2 x (64 KiB
// stash + the 8 KiB layer table) -- still 16 waves per CU, the whole C2 batch resident.
constexpr int gauss_wpb(int LW, int RNG) { return (LW == 3 || RNG != GAUSS_RNG_HBM) ? 8 : 4; }

// UDT ("uniform dt"): every chain integrates with the kernel argument `timestep` and no
// adaption runs in the launch (the host picks it: gauss_uniform_dt).  The step size then
// lives in a scalar register pair instead of a vector pair per lane; same arithmetic, same
// bits, ~3 % shorter launches at the C2 shape (profiles/r04_u_ab.txt).
template <int TMAX, bool REGULAR, bool UNIT, bool FMA, int LW, int RNG = GAUSS_RNG_HBM, bool UDT = false>
__global__ void __launch_bounds__(64 * gauss_wpb(LW, RNG))
hmc_gauss_persist_kernel(const GaussNArgs a)
{
    constexpr int WPB = gauss_wpb(LW, RNG);          // waves per workgroup
    constexpr int WPC = 1 << LW;                     // waves per chain
    __shared__ double xch[WPB];
    __shared__ double ubc[WPB];                      // the chain's acceptance draw, wave to wave
    constexpr int GS = (TMAX % 8 == 0) ? 8 : ((TMAX % 4 == 0) ? 4 : TMAX);   // measured: 8 beats 4 and 16
    constexpr int NG = TMAX / GS;
    __shared__ double stash[RNG == GAUSS_RNG_DUMP ? 1 : WPB][RNG == GAUSS_RNG_DUMP ? 1 : TMAX][64];
    __shared__ double zx[RNG == GAUSS_RNG_HBM ? 1 : XZIG_C + 1];
    if (RNG != GAUSS_RNG_HBM) {
        xzig_load_table(zx, threadIdx.x, WPB * 64);
        __syncthreads();
    }

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

    double dt = (UDT || !a.dt_chain) ? a.timestep : a.dt_chain[chain];
    double uu = 0.0;
    if (RNG == GAUSS_RNG_HBM) uu = a.u[chain];
    // the lane's random stream: identified by (GLOBAL chain, leaf, accumulator),
    // i.e. by WHICH elements the lane owns, not by where it runs (nor by which
    // rank's shard the chain sits in: chain_offset); the redundant groups of a
    // ragged tree share the stream of the leaf they recompute
    // Transition s of a launch draws from stream position rng_offset + s, exactly what
    // a single-transition launch at that position draws: sample_n(n) and n sample()
    // calls see the same chains (as on the long-chain path, hmc_gauss_big.hip).
    Xo128 gen = {0u, 0u, 0u, 0u};
    uint64_t gen_stream = 0;
    if (RNG != GAUSS_RNG_HBM) {
        const int cgrp = grp & ~((1 << (H - leafdepth)) - 1);
        gen_stream = (uint64_t)(chain + a.chain_offset) * (uint64_t)(8 << H) + (uint64_t)(cgrp * 8 + j);
    }
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
        q[t] = (m && RNG != GAUSS_RNG_DUMP) ? a.q0[base + 8 * t] : 0.0;
    }
    if (RNG == GAUSS_RNG_HBM) {
#pragma unroll
        for (int i = 0; i < GS; ++i) {
            const bool m = REGULAR || (8 * i + j < n);
            pa[i] = m ? a.p0[base + 8 * i] : 0.0;
        }
    }
#pragma unroll
    for (int t = GS; t < TMAX; ++t) {
        const bool m = REGULAR || (8 * t + j < n);
        q[t] = (m && RNG != GAUSS_RNG_DUMP) ? a.q0[base + 8 * t] : 0.0;
    }

    const double c_lp = -0.5 * a.k;
    // np.sum((q - x0)**2) of the CURRENT state, carried across transitions
    // (for the start state it is summed group by group inside the first
    // transition, so that the first trajectories need not wait for all of q0)
    LaneSum s0 = {0.0, 0.0};
    double Sq_state = 0.0;
    int64_t nacc = 0;

    // Where a rejected chain gets its old state back from.  When EVERY state is recorded
    // (thin == 1) the state before transition s is what this very lane wrote to the
    // record of transition s - 1 (q0 for s = 0): it is read back from there on the rare
    // rejection, and the per-transition copy of the state to LDS (16 ds_write_b64 per lane)
    // is not made at all; a single transition (n == 1) reads q0 again.  Regular trees only
    // (every lane owns what it reads back).
    const bool stash_lds = !(REGULAR && ((a.samples && a.thin == 1) || a.n == 1) && !a.force_lds_stash);

    for (int s = 0; s < a.n; ++s) {
        const double hdt = 0.5 * dt;
        if (RNG != GAUSS_RNG_HBM) gen = xo_seed(gen_stream, a.rng_seed, a.rng_offset + (uint64_t)s);
        // state before the transition -> LDS (read back only on rejection)
        if (RNG != GAUSS_RNG_DUMP && stash_lds) {
#pragma unroll
            for (int t = 0; t < TMAX; ++t) stash[wib][t][lane] = q[t];
        }

        // Prefetches are issued UNCONDITIONALLY (on the last transition they
        // re-read this transition's data and are ignored): a load under a
        // branch makes the compiler's vmcnt bookkeeping assume it may not have
        // been issued, and the next counted wait then also waits for it.
        const bool more = s + 1 < a.n;
        const double *pc = a.p0 + (int64_t)s * CD + base;
        const double *pn = more ? pc + CD : pc;
        double un = 0.0;
        if (RNG == GAUSS_RNG_HBM) un = a.u[(int64_t)(more ? s + 1 : s) * a.C + chain];

        LaneSum spb = {0.0, 0.0}, sqa = {0.0, 0.0}, spa = {0.0, 0.0};
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            double(&cur)[GS] = (RNG == GAUSS_RNG_HBM && (g & 1)) ? pb : pa;
            double(&nxt)[GS] = (RNG == GAUSS_RNG_HBM && (g & 1)) ? pa : pb;
            if (RNG != GAUSS_RNG_HBM) {
                // this group's momentum draw, np.random.normal (hmc.py:146)
                unsigned want = 0;
#pragma unroll
                for (int i = 0; i < GS; ++i)
                    if (REGULAR || (8 * (g * GS + i) + j < n)) want |= 1u << i;
                xzig_normals<GS>(cur, want, gen, zx);
                if (RNG == GAUSS_RNG_DUMP) {
                    if (cvalid && canonical) {
                        double *go = a.p_dump + (int64_t)s * CD + base;
#pragma unroll
                        for (int i = 0; i < GS; ++i)
                            if (want & (1u << i)) go[8 * (g * GS + i)] = cur[i];
                    }
                    continue;
                }
            } else if (g + 1 < NG) {
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
            (void)nxt;
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
        if (RNG != GAUSS_RNG_HBM) {
            // the acceptance draw, np.random.uniform (hmc.py:151): every lane
            // advances its stream, the chain uses the one of its first lane
            const double ud = xo_uniform53(gen);
            if (LW == 0) {
                uu = shfl_f64(ud, lane & ~((1 << lg) - 1));
            } else {
                // the first lane of the chain's first wave draws for all its waves
                if (wchain == 0 && lane == 0) ubc[wib] = ud;
                __syncthreads();
                uu = ubc[wib & ~(WPC - 1)];
                __syncthreads();
            }
            if (RNG == GAUSS_RNG_DUMP) {
                if (cvalid && writer) a.u_dump[(int64_t)s * a.C + chain] = uu;
                continue;
            }
        }
        if (RNG == GAUSS_RNG_HBM && (NG & 1)) {
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

        if (!UDT && s < a.n_adapt)                            // hmc.py:188-191
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
        } else if (stash_lds) {
#pragma unroll
            for (int t = 0; t < TMAX; ++t) q[t] = stash[wib][t][lane];
        } else {
            const double *prev = (s == 0) ? a.q0 + base
                                          : a.samples + (int64_t)(s - 1) * CD + base;
#pragma unroll
            for (int t = 0; t < TMAX; ++t) q[t] = prev[8 * t];
        }
        if (a.samples && (s + 1) % a.thin == 0 && cvalid && canonical) {
            double *go = a.samples + (int64_t)((s + 1) / a.thin - 1) * CD + base;
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (REGULAR || (8 * t + j < n)) go[8 * t] = q[t];
        }
        if (RNG == GAUSS_RNG_HBM) uu = un;
    }
    if (RNG == GAUSS_RNG_DUMP) return;

    if (cvalid && writer) {
        if (a.n_accepted) a.n_accepted[chain] += nacc;
        if (!UDT && a.n_adapt > 0 && a.dt_chain) a.dt_chain[chain] = dt;
    }
    if (cvalid && canonical) {
        double *go = a.q_out + base;
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (REGULAR || (8 * t + j < n)) go[8 * t] = q[t];
    }
}


}  // namespace binf

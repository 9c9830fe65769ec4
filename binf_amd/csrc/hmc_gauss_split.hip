// Fused Gaussian HMC for FEW chains of D in {768, 1024}: a chain is spread over
// SPLIT = 2 or 4 waves instead of one (the strong-scaling share
// of 4096 chains over 8 GPUs is 512 chains = 512 waves on 1024 SIMDs).  Same
// arithmetic, same bits as hmc_gauss_persist_kernel (hmc_gauss.hip); replaces
// the same reference lines (binf/samplers/hmc.py:92-164,183-191).
//
// Lane (leaf g, accumulator j) of wave `part` owns the elements
// off_g + 8 t + j for t in [part * TS, (part + 1) * TS), TS = TMAX / SPLIT: a
// contiguous piece of numpy's j-th accumulator chain of that leaf.  The
// trajectory is elementwise, so the parts integrate independently; the three
// energy sums are strictly serial along t, so each part adds its TS squares to
// the running sums it receives from the part before (through LDS, one barrier
// per hand-over) and the LAST part finishes them across lanes (xor-shuffles,
// as in the one-wave kernel), decides the accept test and publishes the
// verdict.  SPLIT barriers per transition against 4x / 2x fewer serial FP64
// instructions per wave.
#include "gauss_common.hpp"

namespace binf {

template <int TMAX, int SPLIT, bool UNIT, bool FMA>
__global__ void __launch_bounds__(256) hmc_gauss_split_kernel(const GaussNArgs a)
{
    constexpr int TS = TMAX / SPLIT;                 // elements per lane
    constexpr int CPB = 4 / SPLIT;                   // chains per workgroup
    static_assert(TMAX % SPLIT == 0 && 4 % SPLIT == 0, "bad split");
    __shared__ double stash[4][TS][64];
    __shared__ double hand[CPB][3][64];
    __shared__ double verdict[CPB];

    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int part = wib % SPLIT;
    const int cib = wib / SPLIT;
    const bool last = (part == SPLIT - 1);
    const int j = lane & 7, grp = lane >> 3;         // H == 3: 8 leaves x 8 accumulators
    const int64_t raw = (int64_t)blockIdx.x * CPB + cib;
    const bool cvalid = raw < a.C;
    const int64_t chain = cvalid ? raw : a.C - 1;
    const int64_t CD = a.C * (int64_t)a.D;
    const int64_t base = chain * (int64_t)a.D + grp * (8 * TMAX) + 8 * (part * TS) + j;

    double dt = a.dt_chain ? a.dt_chain[chain] : a.timestep;
    double uu = a.u[chain];
    double q[TS], p[TS], pn[TS];
#pragma unroll
    for (int t = 0; t < TS; ++t) {
        q[t] = a.q0[base + 8 * t];
        p[t] = a.p0[base + 8 * t];
    }

    // Serial part of np.sum along an accumulator chain, handed from part to
    // part; returns the finished sums on the last part (garbage elsewhere).
    // Every wave of the workgroup executes the same barriers.
    double fin[3];
    auto chain_sums = [&](const double(&vb)[TS], const double(&vq)[TS], const double(&vp)[TS]) {
#pragma unroll
        for (int k = 0; k < SPLIT; ++k) {
            if (part == k) {
                double sb = (k == 0) ? vb[0] : hand[cib][0][lane] + vb[0];
                double sq = (k == 0) ? vq[0] : hand[cib][1][lane] + vq[0];
                double sp = (k == 0) ? vp[0] : hand[cib][2][lane] + vp[0];
#pragma unroll
                for (int t = 1; t < TS; ++t) {
                    sb = sb + vb[t];
                    sq = sq + vq[t];
                    sp = sp + vp[t];
                }
                if (k < SPLIT - 1) {
                    hand[cib][0][lane] = sb;
                    hand[cib][1][lane] = sq;
                    hand[cib][2][lane] = sp;
                } else {
                    const LaneSum lb = {sb, 0.0}, lq = {sq, 0.0}, lp = {sp, 0.0};
                    fin[0] = chain_sum_finish<true>(lb, TMAX, 0, lane, 3, 3);
                    fin[1] = chain_sum_finish<true>(lq, TMAX, 0, lane, 3, 3);
                    fin[2] = chain_sum_finish<true>(lp, TMAX, 0, lane, 3, 3);
                }
            }
            if (k < SPLIT - 1) __syncthreads();
        }
    };

    const double c_lp = -0.5 * a.k;
    double v0[TS], v1[TS], v2[TS];
#pragma unroll
    for (int t = 0; t < TS; ++t) {
        const double d = UNIT ? q[t] : q[t] - a.x0;
        v0[t] = 0.0;
        v1[t] = d * d;
        v2[t] = 0.0;
    }
    chain_sums(v0, v1, v2);
    double Sq_state = fin[1];                        // meaningful on the last part only
    int64_t nacc = 0;
    // the last part's reads of `hand` above must complete before part 0 writes
    // it again in transition 0 (inside the loop the verdict barrier does this)
    __syncthreads();

    for (int s = 0; s < a.n; ++s) {
        const double hdt = 0.5 * dt;
#pragma unroll
        for (int t = 0; t < TS; ++t) stash[wib][t][lane] = q[t];
        // next transition's draw, fetched unconditionally (see hmc_gauss.hip)
        const bool more = s + 1 < a.n;
        const double *pnext = a.p0 + (int64_t)(more ? s + 1 : s) * CD + base;
#pragma unroll
        for (int t = 0; t < TS; ++t) pn[t] = pnext[8 * t];
        const double un = a.u[(int64_t)(more ? s + 1 : s) * a.C + chain];

#pragma unroll
        for (int t = 0; t < TS; ++t) v0[t] = p[t] * p[t];                      // hmc.py:148
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TS; ++t)                                          // hmc.py:116
            p[t] = kick<FMA>(p[t], hdt, gauss_grad<UNIT>(q[t], a.k, a.x0));
        for (int l = 0; l < a.nsteps - 1; ++l) {                              // hmc.py:118-120
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                q[t] = drift<FMA>(q[t], p[t], dt);
                p[t] = kick<FMA>(p[t], dt, gauss_grad<UNIT>(q[t], a.k, a.x0));
            }
        }
#pragma unroll
        for (int t = 0; t < TS; ++t) {                                        // hmc.py:122-123
            q[t] = drift<FMA>(q[t], p[t], dt);
            p[t] = kick<FMA>(p[t], hdt, gauss_grad<UNIT>(q[t], a.k, a.x0));
        }
#pragma unroll
        for (int t = 0; t < TS; ++t) {                                        // hmc.py:150
            const double d = UNIT ? q[t] : q[t] - a.x0;
            v1[t] = d * d;
            v2[t] = p[t] * p[t];
        }
        chain_sums(v0, v1, v2);

        if (last) {
            const double Eb = -(c_lp * Sq_state) + 0.5 * fin[0];
            const double Ea = -(c_lp * fin[1]) + 0.5 * fin[2];
            double x = -(Ea - Eb);                                            // hmc.py:151
            x = (x < -308.0) ? -308.0 : x;
            x = (x > 709.0) ? 709.0 : x;
            const bool ok = uu < exp_clipped_range(x);
            if (ok) Sq_state = fin[1];
            if (lane == 0) {
                verdict[cib] = ok ? 1.0 : 0.0;
                if (cvalid) {
                    const int64_t o = (int64_t)s * a.C + chain;
                    if (a.accepted) a.accepted[o] = ok ? 1 : 0;
                    if (a.e_before) a.e_before[o] = Eb;
                    if (a.e_after) a.e_after[o] = Ea;
                }
            }
        }
        __syncthreads();
        const bool acc = verdict[cib] != 0.0;
        if (s < a.n_adapt)                                                    // hmc.py:188-191
            dt = acc ? dt * a.uprate : dt * a.downrate;
        if (acc) {
            nacc += 1;
        } else {
#pragma unroll
            for (int t = 0; t < TS; ++t) q[t] = stash[wib][t][lane];
        }
        if (a.samples && (s + 1) % a.thin == 0 && cvalid) {
            double *go = a.samples + (int64_t)((s + 1) / a.thin - 1) * CD + base;
#pragma unroll
            for (int t = 0; t < TS; ++t) go[8 * t] = q[t];
        }
#pragma unroll
        for (int t = 0; t < TS; ++t) p[t] = pn[t];
        uu = un;
        // the verdict slot is rewritten only after the next transition's
        // hand-over barriers (SPLIT >= 2), so no extra barrier is needed here
    }

    if (cvalid && last && lane == 0) {
        if (a.n_accepted) a.n_accepted[chain] += nacc;
        if (a.n_adapt > 0 && a.dt_chain) a.dt_chain[chain] = dt;
    }
    if (cvalid) {
        double *go = a.q_out + base;
#pragma unroll
        for (int t = 0; t < TS; ++t) go[8 * t] = q[t];
    }
}

template <int TMAX, int SPLIT>
static hipError_t launch_split_ts(const GaussNArgs &a, bool unit, bool fma, hipStream_t st)
{
    constexpr int CPB = 4 / SPLIT;
    const dim3 grid((unsigned)((a.C + CPB - 1) / CPB));
    if (unit) {
        if (fma) hmc_gauss_split_kernel<TMAX, SPLIT, true, true><<<grid, 256, 0, st>>>(a);
        else     hmc_gauss_split_kernel<TMAX, SPLIT, true, false><<<grid, 256, 0, st>>>(a);
    } else {
        if (fma) hmc_gauss_split_kernel<TMAX, SPLIT, false, true><<<grid, 256, 0, st>>>(a);
        else     hmc_gauss_split_kernel<TMAX, SPLIT, false, false><<<grid, 256, 0, st>>>(a);
    }
    return hipGetLastError();
}

int gauss_split_factor(int64_t C, int32_t H, bool regular, int tneed)
{
    static int forced = -1;
    if (forced < 0) {
        const char *e = getenv("BINF_GAUSS_SPLIT");          // development aid: 1, 2, 4
        forced = e ? atoi(e) : 0;
    }
    if (H != 3 || !regular || (tneed != 16 && tneed != 12)) return 1;
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    // one wave per chain fills the 1024 SIMDs four deep at 4096 chains
    if (C <= 1024) return 4;
    if (C <= 2048) return 2;
    return 1;
}

hipError_t launch_gauss_split(const GaussNArgs &a, int tneed, int split, bool unit, bool fma,
                              hipStream_t st)
{
    if (tneed == 16) {
        return split == 4 ? launch_split_ts<16, 4>(a, unit, fma, st)
                          : launch_split_ts<16, 2>(a, unit, fma, st);
    }
    return split == 4 ? launch_split_ts<12, 4>(a, unit, fma, st)
                      : launch_split_ts<12, 2>(a, unit, fma, st);
}

}  // namespace binf

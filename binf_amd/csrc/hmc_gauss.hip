// Fused HMC transition on an isotropic Gaussian, one launch = one
// HMCSampler.sample() for every chain.  gfx950 (MI355X), wave64.
//
// Replaces (reference paths): binf/samplers/hmc.py:92-164,183-191 and the
// TestHO log_prob/gradient of binf/pdf/__init__.py:181-191.
//
// Mapping.  A chain's D coordinates are owned by G = 8 * 2^H lanes of ONE wave
// (H = height of numpy's pairwise-sum tree for length D), so 64/G chains share
// a wave.  Lane (leaf path g, accumulator j) owns elements off_g + 8t + j,
// t = 0..TMAX-1 -- exactly the elements numpy's j-th strided accumulator of
// that leaf adds up, in order.  The whole trajectory (q, p) stays in VGPRs;
// HBM sees q0, p0 once in and q_out once out (8-byte accesses in 64-byte
// segments: measured at the same 6.3 TB/s as 16-byte coalesced streaming).
//
// Schedule.  The Gaussian is separable, so the nsteps-long trajectory of a
// group of GS elements per lane is run to completion as soon as that group's
// loads have landed, while the later loads of the wave are still in flight
// ("element-group-major" order); the in-lane partial sums of the four energy
// reductions are carried along in t order.  A wave owns NCH chain slots whose
// loads are all issued up front, so the stores of slot 0 overlap the compute
// of slot 1.  Energy reductions finish with xor-shuffles 1,2,4 inside the
// leaf (numpy's ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))), the leaf's tail
// elements, then xor-shuffles 8,16,32 up the leaf tree: bit-identical to
// np.sum.
#include "gauss_common.hpp"

namespace binf {

struct GaussArgs {
    const double *q0;
    const double *p0;
    const double *u;
    double *q_out;
    uint8_t *accepted;
    int64_t *n_accepted;
    double *e_before;
    double *e_after;
    double *dt_chain;
    double timestep;
    double k;
    double x0;
    double uprate;
    double downrate;
    int64_t C;
    int32_t D;
    int32_t nsteps;
    int32_t H;      // tree height, G = 8 << H lanes per chain
    int32_t adapt;
};

template <int TMAX, bool REGULAR, bool UNIT, bool FMA, int NCH>
__global__ void __launch_bounds__(256)
hmc_gauss_wave_kernel(const GaussArgs a)
{
    constexpr int GS = (TMAX % 4 == 0) ? 4 : TMAX;   // elements per group
    constexpr int NG = TMAX / GS;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int H = a.H;
    const int lg = 3 + H;                  // log2(lanes per chain)
    const int slot = lane & ((1 << lg) - 1);
    const int j = slot & 7;

    int off, n, leafdepth, canonical;
    if (REGULAR) {
        n = 8 * TMAX;
        off = (slot >> 3) * n;
        leafdepth = H;
        canonical = 1;
    } else {
        const Leaf L = pairwise_leaf(a.D, H, slot >> 3);
        off = L.off;
        n = L.len;
        leafdepth = L.depth;
        canonical = L.canonical;
    }
    const int T = (n >= 8) ? (n >> 3) : 0;
    const int rem = (n >= 8) ? (n & 7) : n;

    // ---- issue every load of every chain slot up front ---------------------
    int64_t chain[NCH];
    bool cvalid[NCH];
    double dtv[NCH], uv[NCH];
    double q[NCH][TMAX], p[NCH][TMAX];
    // per-chain scalars first: loads return in order, so a scalar issued
    // behind the bulk loads would make its first use wait for all of them
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int64_t raw = ((wave * NCH + c) << (6 - lg)) + (lane >> lg);
        cvalid[c] = raw < a.C;
        chain[c] = cvalid[c] ? raw : a.C - 1;
        dtv[c] = a.dt_chain ? a.dt_chain[chain[c]] : a.timestep;
        uv[c] = a.u[chain[c]];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int64_t base = chain[c] * (int64_t)a.D + off + j;
        const double *gq = a.q0 + base;
        const double *gp = a.p0 + base;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            const bool m = REGULAR || (8 * t + j < n);
            q[c][t] = m ? gq[8 * t] : 0.0;
            p[c][t] = m ? gp[8 * t] : 0.0;
        }
    }

    const double c_lp = -0.5 * a.k;           // "-0.5 * k", pdf/__init__.py:185

#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const double dt = dtv[c];
        const double hdt = 0.5 * dt;          // "0.5 * timestep" formed first
        const double uu = uv[c];

        LaneSum sqb = {0.0, 0.0}, spb = {0.0, 0.0};   // E_before parts
        LaneSum sqa = {0.0, 0.0}, spa = {0.0, 0.0};   // E_after parts
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            // pin this group's values here (see hmc_gauss_persist.hip)
#pragma unroll
            for (int i = 0; i < GS; ++i) {
                const int t = g * GS + i;
                asm volatile("" : "+v"(p[c][t]), "+v"(q[c][t]));
            }
            // E_before terms: (q-x0)**2, p**2                   hmc.py:148
#pragma unroll
            for (int i = 0; i < GS; ++i) {
                const int t = g * GS + i;
                const double d = UNIT ? q[c][t] : q[c][t] - a.x0;
                lane_sum_add<REGULAR>(sqb, d * d, t, T);
                lane_sum_add<REGULAR>(spb, p[c][t] * p[c][t], t, T);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep q0**2 out of the loop's way
            // _leapfrog                                         hmc.py:116-123
#pragma unroll
            for (int i = 0; i < GS; ++i) {
                const int t = g * GS + i;
                p[c][t] = kick<FMA>(p[c][t], hdt, gauss_grad<UNIT>(q[c][t], a.k, a.x0));
            }
            for (int s = 0; s < a.nsteps - 1; ++s) {
#pragma unroll
                for (int i = 0; i < GS; ++i) {
                    const int t = g * GS + i;
                    q[c][t] = drift<FMA>(q[c][t], p[c][t], dt);
                    p[c][t] = kick<FMA>(p[c][t], dt, gauss_grad<UNIT>(q[c][t], a.k, a.x0));
                }
            }
#pragma unroll
            for (int i = 0; i < GS; ++i) {
                const int t = g * GS + i;
                q[c][t] = drift<FMA>(q[c][t], p[c][t], dt);
                p[c][t] = kick<FMA>(p[c][t], hdt, gauss_grad<UNIT>(q[c][t], a.k, a.x0));
            }
            // E_after terms                                     hmc.py:150
#pragma unroll
            for (int i = 0; i < GS; ++i) {
                const int t = g * GS + i;
                const double d = UNIT ? q[c][t] : q[c][t] - a.x0;
                lane_sum_add<REGULAR>(sqa, d * d, t, T);
                lane_sum_add<REGULAR>(spa, p[c][t] * p[c][t], t, T);
            }
        }
        const double Sqb = chain_sum_finish<REGULAR>(sqb, T, rem, lane, H, leafdepth);
        const double Spb = chain_sum_finish<REGULAR>(spb, T, rem, lane, H, leafdepth);
        const double Sqa = chain_sum_finish<REGULAR>(sqa, T, rem, lane, H, leafdepth);
        const double Spa = chain_sum_finish<REGULAR>(spa, T, rem, lane, H, leafdepth);
        const double Eb = -(c_lp * Sqb) + 0.5 * Spb;
        const double Ea = -(c_lp * Sqa) + 0.5 * Spa;

        // acc = uniform < exp(-(E_after - E_before)), csb clipped exp  hmc.py:151
        double x = -(Ea - Eb);
        x = (x < -308.0) ? -308.0 : x;
        x = (x > 709.0) ? 709.0 : x;      // NaN falls through both, as np.clip
        const bool acc = uu < exp(x);

        if (cvalid[c] && slot == 0) {
            a.accepted[chain[c]] = acc ? 1 : 0;
            if (a.n_accepted && acc) a.n_accepted[chain[c]] += 1;   // hmc.py:161
            if (a.e_before) a.e_before[chain[c]] = Eb;
            if (a.e_after) a.e_after[chain[c]] = Ea;
            if (a.adapt)                                     // hmc.py:188-191
                a.dt_chain[chain[c]] = acc ? dt * a.uprate : dt * a.downrate;
        }

        // return value: the proposal if accepted, else the old state  hmc.py:159-164
        const int64_t base = chain[c] * (int64_t)a.D + off + j;
        double *go = a.q_out + base;
        const double *gq = a.q0 + base;
        if (cvalid[c] && canonical) {
            if (acc) {
#pragma unroll
                for (int t = 0; t < TMAX; ++t)
                    if (REGULAR || (8 * t + j < n)) go[8 * t] = q[c][t];
            } else if (a.q_out != a.q0) {
#pragma unroll
                for (int t = 0; t < TMAX; ++t)
                    if (REGULAR || (8 * t + j < n)) go[8 * t] = gq[8 * t];
            }
        }
    }
}

template <int TMAX, bool REGULAR, int NCH>
static hipError_t launch_trn(const GaussArgs &a, bool unit, bool fma,
                             dim3 grid, hipStream_t st)
{
    if (unit) {
        if (fma) hmc_gauss_wave_kernel<TMAX, REGULAR, true, true, NCH><<<grid, 256, 0, st>>>(a);
        else     hmc_gauss_wave_kernel<TMAX, REGULAR, true, false, NCH><<<grid, 256, 0, st>>>(a);
    } else {
        if (fma) hmc_gauss_wave_kernel<TMAX, REGULAR, false, true, NCH><<<grid, 256, 0, st>>>(a);
        else     hmc_gauss_wave_kernel<TMAX, REGULAR, false, false, NCH><<<grid, 256, 0, st>>>(a);
    }
    return hipGetLastError();
}

template <int TMAX>
static hipError_t launch_t(const GaussArgs &a, bool regular, int nch, bool unit,
                           bool fma, dim3 grid, hipStream_t st)
{
    if (nch == 2)
        return regular ? launch_trn<TMAX, true, 2>(a, unit, fma, grid, st)
                       : launch_trn<TMAX, false, 2>(a, unit, fma, grid, st);
    return regular ? launch_trn<TMAX, true, 1>(a, unit, fma, grid, st)
                   : launch_trn<TMAX, false, 1>(a, unit, fma, grid, st);
}

}  // namespace binf

using namespace binf;

// Number of chain slots per wave.  Measured on MI355X at C2 (4096 chains,
// D = 1024, L = 20, random data): NCH = 1 25.0 us, NCH = 2 29.1 us per launch
// (profiles/r01_b_notes.md), so 1 is the default; BINF_GAUSS_NCH=2 selects
// the two-slot variant for experiments.
static int pick_nch(int64_t waves1, int tneed)
{
    static int forced = -1;
    if (forced < 0) {
        const char *e = getenv("BINF_GAUSS_NCH");
        forced = e ? atoi(e) : 0;
    }
    (void)waves1;
    (void)tneed;
    return forced == 2 ? 2 : 1;
}

extern "C" int32_t binf_hmc_sample_gauss_f64(
    const double *q0, const double *p0, const double *u, double *q_out,
    uint8_t *accepted, int64_t *n_accepted, double *e_before, double *e_after,
    double timestep, double *dt_chain, int64_t C, int64_t D, int32_t nsteps,
    double k, double x0, int32_t adapt, double uprate, double downrate,
    int32_t mode, void *stream)
{
    if (C < 0 || D < 1 || nsteps < 1)
        return fail(BINF_E_ARG, "hmc_sample_gauss: need C>=0, D>=1, nsteps>=1 (C=%lld D=%lld nsteps=%d)",
                    (long long)C, (long long)D, nsteps);
    if (mode != BINF_MODE_EXACT && mode != BINF_MODE_FMA)
        return fail(BINF_E_ARG, "hmc_sample_gauss: unknown mode %d", mode);
    if (C == 0) return 0;
    if (!q0 || !p0 || !u || !q_out || !accepted)
        return fail(BINF_E_ARG, "hmc_sample_gauss: null buffer");
    if (adapt && !dt_chain)
        return fail(BINF_E_ARG, "hmc_sample_gauss: adapt needs dt_chain");
    const int64_t bytes = C * D * (int64_t)sizeof(double);
    const char *qo = (const char *)q_out, *qi = (const char *)q0, *pi = (const char *)p0;
    if ((qo != qi && qo < qi + bytes && qi < qo + bytes) ||
        (qo < pi + bytes && pi < qo + bytes))
        return fail(BINF_E_ALIAS, "hmc_sample_gauss: q_out overlaps q0/p0 (only q_out == q0 is allowed)");
    if (D > 1024)
        return fail(BINF_E_UNSUPPORTED, "hmc_sample_gauss: D=%lld > 1024 not covered by the fused kernel", (long long)D);
    const int32_t H = pairwise_tree_height(D);
    if (H > 3)
        return fail(BINF_E_UNSUPPORTED, "hmc_sample_gauss: pairwise tree height %d > 3 for D=%lld", H, (long long)D);

    // widest leaf decides how many elements a lane owns
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
    GaussArgs a;
    a.q0 = q0; a.p0 = p0; a.u = u; a.q_out = q_out; a.accepted = accepted;
    a.n_accepted = n_accepted;
    a.e_before = e_before; a.e_after = e_after; a.dt_chain = dt_chain;
    a.timestep = timestep; a.k = k; a.x0 = x0; a.uprate = uprate;
    a.downrate = downrate; a.C = C; a.D = (int32_t)D; a.nsteps = nsteps;
    a.H = H; a.adapt = adapt;

    const int64_t chains_per_wave = 64 >> (3 + H);
    const int64_t waves1 = (C + chains_per_wave - 1) / chains_per_wave;
    const int nch = pick_nch(waves1, tneed);
    const int64_t waves = (waves1 + nch - 1) / nch;
    const int64_t blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffffLL)
        return fail(BINF_E_UNSUPPORTED, "hmc_sample_gauss: too many chains");
    dim3 grid((unsigned)blocks);
    hipStream_t st = (hipStream_t)stream;
    const bool unit = (k == 1.0 && x0 == 0.0);
    const bool fma = (mode == BINF_MODE_FMA);
    hipError_t e;
    if (tneed <= 1)       e = launch_t<1>(a, regular && tneed == 1, nch, unit, fma, grid, st);
    else if (tneed <= 2)  e = launch_t<2>(a, regular && tneed == 2, nch, unit, fma, grid, st);
    else if (tneed <= 4)  e = launch_t<4>(a, regular && tneed == 4, nch, unit, fma, grid, st);
    else if (tneed <= 8)  e = launch_t<8>(a, regular && tneed == 8, nch, unit, fma, grid, st);
    else if (tneed <= 12) e = launch_t<12>(a, regular && tneed == 12, nch, unit, fma, grid, st);
    else                  e = launch_t<16>(a, regular && tneed == 16, nch, unit, fma, grid, st);
    if (e != hipSuccess) return hip_fail(e, "hmc_gauss_wave_kernel launch");
    return 0;
}

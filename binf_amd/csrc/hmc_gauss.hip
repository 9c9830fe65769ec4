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
#include "hmc_gauss_kernel.hpp"

namespace binf {

template <int TMAX, bool REGULAR, int LW, bool UDT = false>
static hipError_t launch_n_trl(const GaussNArgs &a, bool unit, bool fma, dim3 grid,
                               hipStream_t st)
{
    constexpr int BS = (LW == 3) ? 512 : 256;
    constexpr int R = GAUSS_RNG_HBM;
    if (unit) {
        if (fma) hmc_gauss_persist_kernel<TMAX, REGULAR, true, true, LW, R, UDT><<<grid, BS, 0, st>>>(a);
        else     hmc_gauss_persist_kernel<TMAX, REGULAR, true, false, LW, R, UDT><<<grid, BS, 0, st>>>(a);
    } else {
        if (fma) hmc_gauss_persist_kernel<TMAX, REGULAR, false, true, LW, R, UDT><<<grid, BS, 0, st>>>(a);
        else     hmc_gauss_persist_kernel<TMAX, REGULAR, false, false, LW, R, UDT><<<grid, BS, 0, st>>>(a);
    }
    return hipGetLastError();
}

// One step size for the whole batch and no adaption in this launch: the UDT instantiation
// (hmc_gauss_kernel.hpp) -- built for the regular one-wave-per-chain shapes, where the
// vector registers it frees are worth ~3 %; every other launch reads `timestep` per lane.
static bool gauss_uniform_dt(const GaussNArgs &a)
{
    static int off = -1;
    if (off < 0) {
        const char *e = getenv("BINF_GAUSS_UNIFORM_DT");     // development aid: =0 disables
        off = (e && e[0] == '0') ? 1 : 0;
    }
    return !off && a.dt_chain == nullptr && a.n_adapt == 0;
}

template <int TMAX>
static hipError_t launch_n_t(const GaussNArgs &a, bool regular, bool unit, bool fma,
                             dim3 grid, hipStream_t st)
{
    if (regular && gauss_uniform_dt(a))
        return launch_n_trl<TMAX, true, 0, true>(a, unit, fma, grid, st);
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

// development aid: BINF_GAUSS_STASH=lds keeps the per-transition LDS stash even when every
// state is recorded (A/B of the read-back-from-the-record restore)
int gauss_force_lds_stash()
{
    static int forced = -1;
    if (forced < 0) {
        const char *e = getenv("BINF_GAUSS_STASH");
        forced = (e && e[0] == 'l') ? 1 : 0;
    }
    return forced;
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
    const GaussPlan plan = gauss_plan(C, D);
    const int32_t H = plan.H;
    if (H > 6)
        return fail(BINF_E_UNSUPPORTED, "hmc_sample_gauss: pairwise tree height %d > 6 for D=%lld", H, (long long)D);
    const int LW = plan.LW;                          // log2(waves per chain)
    const int64_t bytes = C * D * (int64_t)sizeof(double);
    const char *qo = (const char *)q_out, *qi = (const char *)q0, *pi = (const char *)p0;
    if ((qo != qi && qo < qi + bytes && qi < qo + bytes) ||
        (qo < pi + bytes * n && pi < qo + bytes))
        return fail(BINF_E_ALIAS, "hmc_sample_gauss: q_out overlaps q0/p0 (only q_out == q0 is allowed)");

    const int tneed = plan.tneed;
    const bool regular = plan.regular;
    GaussNArgs a;
    a.q0 = q0; a.p0 = p0; a.u = u; a.q_out = q_out; a.samples = samples;
    a.accepted = accepted; a.n_accepted = n_accepted; a.e_before = e_before;
    a.e_after = e_after; a.dt_chain = dt_chain; a.timestep = timestep; a.k = k;
    a.x0 = x0; a.uprate = uprate; a.downrate = downrate; a.C = C;
    a.D = (int32_t)D; a.nsteps = nsteps; a.H = H; a.n = n; a.thin = thin;
    a.n_adapt = n_adapt < n ? n_adapt : n;
    a.stagger = gauss_stagger(n);
    a.force_lds_stash = gauss_force_lds_stash();
    a.rng_seed = 0; a.rng_offset = 0; a.chain_offset = 0; a.p_dump = nullptr; a.u_dump = nullptr;

    const int64_t blocks = plan.blocks;
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

// How many waves binf_hmc_sample_[n_]gauss_f64 spreads a chain over for a batch
// of C chains of length D: 1 (one-wave chains; also for every shape the split
// kernel does not serve), 2 or 4 (few chains of D = 768 / 1024), or 2 / 4 / 8 for
// D > 1024.  Informational: lets a caller pick the draw source that suits the
// launch shape (binf_amd/samplers/hmc.py).
extern "C" int32_t binf_hmc_gauss_waves_per_chain(int64_t C, int64_t D)
{
    if (C < 1 || D < 1 || D > 8192) return 0;
    const GaussPlan p = gauss_plan(C, D);
    if (p.H > 6) return 0;
    if (p.LW > 0) return 1 << p.LW;
    return gauss_split_factor(C, p.H, p.regular, p.tneed);
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

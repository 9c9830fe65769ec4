// Fused Gaussian HMC with the random draws generated INSIDE the sampling kernel:
// HMCSampler.sample() as the reference defines it -- np.random.normal(size=
// q.shape), the trajectory, np.random.uniform() (binf/samplers/hmc.py:146-151)
// -- n transitions per launch, with no momentum buffer in HBM at all.  Same
// kernel template as hmc_gauss.hip (hmc_gauss_kernel.hpp, RNG = 1); the draws
// come from per-lane xoshiro128++ streams + a 1024-layer ziggurat (xoshiro.hpp).
//
// binf_hmc_gauss_rng_draws_f64 runs the same template with RNG = 2: it writes
// the draws the fused kernel WOULD consume for (seed, offset, C, D, n) and
// integrates nothing, so that
//     sample_n_rng(seed, offset)  ==  sample_n(p0 = draws, u = draws)   bit for bit
// (tests/test_gpu_rng.py); the statistical quality of the stream is tested on the
// dump.  Same shapes as the persistent kernel (D <= 8192, tree height <= 6; chains
// of 2 / 4 / 8 waves pass the acceptance draw from wave to wave through LDS);
// longer chains use the stand-alone generator kernels (rng.hip).
#include "hmc_gauss_rng_launch.hpp"

namespace binf {

template <int RNG>
static hipError_t launch_rng(const GaussNArgs &a, const GaussPlan &p, bool unit, bool fma,
                             hipStream_t st)
{
    if (p.LW > 0) return launch_gauss_rng_wide(a, p, RNG, unit, fma, st);   // hmc_gauss_rng_wide.hip
    const dim3 grid((unsigned)((p.blocks + 1) / 2));   // 8 waves per workgroup (gauss_wpb)
    const int t = p.tneed;
    // one step size for the batch, no adaption: the scalar-register instantiation (UDT,
    // hmc_gauss_kernel.hpp), as in hmc_gauss.hip
    const bool udt = RNG == GAUSS_RNG_FUSED && !a.dt_chain && a.n_adapt == 0;
#define BINF_RNG_CASE(T)                                                            \
    if (p.regular && t == T && udt)                                                 \
        return launch_rng_tr<T, true, RNG, 0, RNG == GAUSS_RNG_FUSED>(a, unit, fma, grid, st); \
    return (p.regular && t == T) ? launch_rng_tr<T, true, RNG>(a, unit, fma, grid, st) \
                                 : launch_rng_tr<T, false, RNG>(a, unit, fma, grid, st)
    if (t <= 1) { BINF_RNG_CASE(1); }
    if (t <= 2) { BINF_RNG_CASE(2); }
    if (t <= 4) { BINF_RNG_CASE(4); }
    if (t <= 8) { BINF_RNG_CASE(8); }
    if (t <= 12) { BINF_RNG_CASE(12); }
    BINF_RNG_CASE(16);
#undef BINF_RNG_CASE
}

static int32_t rng_plan(const char *what, int64_t C, int64_t D, GaussPlan &p)
{
    if (D > 8192)
        return fail(BINF_E_UNSUPPORTED, "%s: D=%lld > 8192 not covered by the persistent kernel",
                    what, (long long)D);
    p = gauss_plan(C, D);
    if (p.H > 6)
        return fail(BINF_E_UNSUPPORTED, "%s: pairwise tree height %d > 6 for D=%lld", what, p.H,
                    (long long)D);
    if (p.blocks > 0x7fffffffLL) return fail(BINF_E_UNSUPPORTED, "%s: too many chains", what);
    return 0;
}

}  // namespace binf

using namespace binf;

extern "C" int32_t binf_hmc_sample_n_gauss_rng_f64(
    const double *q0, double *q_out, double *samples, uint8_t *accepted,
    int64_t *n_accepted, double *e_before, double *e_after, double timestep,
    double *dt_chain, int64_t C, int64_t D, int32_t nsteps, int32_t n, int32_t thin,
    double k, double x0, int32_t n_adapt, double uprate, double downrate, int32_t mode,
    uint64_t seed, uint64_t offset, int64_t chain_offset, void *stream)
{
    const char *what = "hmc_sample_n_gauss_rng";
    if (C < 0 || D < 1 || nsteps < 1 || n < 1 || thin < 1 || n_adapt < 0 || chain_offset < 0)
        return fail(BINF_E_ARG, "%s: need C>=0, D>=1, nsteps>=1, n>=1, thin>=1, n_adapt>=0, "
                    "chain_offset>=0", what);
    if (mode != BINF_MODE_EXACT && mode != BINF_MODE_FMA)
        return fail(BINF_E_ARG, "%s: unknown mode %d", what, mode);
    if (C == 0) return 0;
    if (!q0 || !q_out) return fail(BINF_E_ARG, "%s: null buffer", what);
    if (n_adapt > 0 && !dt_chain) return fail(BINF_E_ARG, "%s: adaption needs dt_chain", what);
    const int64_t bytes = C * D * (int64_t)sizeof(double);
    const char *qo = (const char *)q_out, *qi = (const char *)q0;
    if (qo != qi && qo < qi + bytes && qi < qo + bytes)
        return fail(BINF_E_ALIAS, "%s: q_out overlaps q0 (only q_out == q0 is allowed)", what);
    GaussPlan p;
    if (int32_t rc = rng_plan(what, C, D, p)) return rc;
    GaussNArgs a;
    a.q0 = q0; a.p0 = nullptr; a.u = nullptr; a.q_out = q_out; a.samples = samples;
    a.accepted = accepted; a.n_accepted = n_accepted; a.e_before = e_before;
    a.e_after = e_after; a.dt_chain = dt_chain; a.timestep = timestep; a.k = k;
    a.x0 = x0; a.uprate = uprate; a.downrate = downrate; a.C = C;
    a.D = (int32_t)D; a.nsteps = nsteps; a.H = p.H; a.n = n; a.thin = thin;
    a.n_adapt = n_adapt < n ? n_adapt : n;
    a.stagger = 0;
    a.force_lds_stash = gauss_force_lds_stash();
    a.rng_seed = seed; a.rng_offset = offset; a.chain_offset = chain_offset;
    a.p_dump = nullptr; a.u_dump = nullptr;
    const hipError_t e = launch_rng<GAUSS_RNG_FUSED>(a, p, k == 1.0 && x0 == 0.0,
                                                    mode == BINF_MODE_FMA, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "hmc_gauss_persist_kernel (fused generator) launch");
    return 0;
}

extern "C" int32_t binf_hmc_gauss_rng_draws_f64(double *p0_out, double *u_out, int64_t C,
                                                int64_t D, int32_t n, uint64_t seed,
                                                uint64_t offset, int64_t chain_offset,
                                                void *stream)
{
    const char *what = "hmc_gauss_rng_draws";
    if (C < 0 || D < 1 || n < 1 || chain_offset < 0)
        return fail(BINF_E_ARG, "%s: need C>=0, D>=1, n>=1, chain_offset>=0", what);
    if (C == 0) return 0;
    if (!p0_out || !u_out) return fail(BINF_E_ARG, "%s: null buffer", what);
    GaussPlan p;
    if (int32_t rc = rng_plan(what, C, D, p)) return rc;
    GaussNArgs a = {};
    a.C = C; a.D = (int32_t)D; a.nsteps = 1; a.H = p.H; a.n = n; a.thin = 1;
    a.k = 1.0;
    a.rng_seed = seed; a.rng_offset = offset; a.chain_offset = chain_offset;
    a.p_dump = p0_out; a.u_dump = u_out;
    const hipError_t e = launch_rng<GAUSS_RNG_DUMP>(a, p, true, false, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "hmc_gauss_persist_kernel (draw dump) launch");
    return 0;
}

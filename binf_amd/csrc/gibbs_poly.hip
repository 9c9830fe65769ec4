// n sweeps of the example's Gibbs loop (example_script.py:33-34 around
// binf/samplers/gibbs.py:136-151) in one launch: the multi-sweep instantiations of
// poly_chain_kernel.hpp.  Contract: include/binf_hip.h, binf_gibbs_poly_sample_n_f64.
#include "poly_chain_kernel.hpp"

namespace binf {

int32_t poly_chain_tcount(int32_t N, int32_t H);      // hmc_poly_wave.hip

template <int KMAX>
static hipError_t launch_gibbs_k(const PolyChainArgs &a, int move, bool fma, hipStream_t st)
{
    const int64_t chains_per_wave = 64 >> (3 + a.H);
    const int64_t waves = (a.C + chains_per_wave - 1) / chains_per_wave;
    const dim3 grid((unsigned)((waves + 3) / 4));
    if (move == POLY_MOVE_RWMC)
        poly_chain_kernel<KMAX, false, true, POLY_MOVE_RWMC><<<grid, 256, 0, st>>>(a);
    else if (fma)
        poly_chain_kernel<KMAX, true, true, POLY_MOVE_HMC><<<grid, 256, 0, st>>>(a);
    else
        poly_chain_kernel<KMAX, false, true, POLY_MOVE_HMC><<<grid, 256, 0, st>>>(a);
    return hipGetLastError();
}

static bool overlap(const void *a, int64_t na, const void *b, int64_t nb)
{
    const char *x = (const char *)a, *y = (const char *)b;
    return a && b && x < y + nb && y < x + na;
}

}  // namespace binf

using namespace binf;

extern "C" int32_t binf_gibbs_poly_sample_n_f64(const binf_gibbs_poly_args *g, void *stream)
{
    if (!g) return fail(BINF_E_ARG, "gibbs_poly: null argument block");
    if (g->struct_size != sizeof(binf_gibbs_poly_args))
        return fail(BINF_E_ARG, "gibbs_poly: struct_size %llu, this library expects %llu",
                    (unsigned long long)g->struct_size,
                    (unsigned long long)sizeof(binf_gibbs_poly_args));
    const int64_t C = g->C, K = g->K, N = g->N;
    if (C < 0 || K < 1 || N < 0 || g->n < 1 || g->thin < 1 || g->chain_offset < 0 || g->n_adapt < 0)
        return fail(BINF_E_ARG, "gibbs_poly: need C>=0, K>=1, N>=0, n>=1, thin>=1, chain_offset>=0");
    if (g->move != BINF_MOVE_HMC && g->move != BINF_MOVE_RWMC)
        return fail(BINF_E_ARG, "gibbs_poly: unknown move %d", g->move);
    if (g->mode != BINF_MODE_EXACT && g->mode != BINF_MODE_FMA)
        return fail(BINF_E_ARG, "gibbs_poly: unknown mode %d", g->mode);
    if (g->move == BINF_MOVE_HMC && g->nsteps < 1)
        return fail(BINF_E_ARG, "gibbs_poly: nsteps >= 1 required");
    if (g->gp_where < 0 || g->gp_where > 2)
        return fail(BINF_E_ARG, "gibbs_poly: gp_where must be 0, 1 or 2");
    if (!g->keep_precision && !(g->gamma_shape > 0.0))
        return fail(BINF_E_ARG, "gibbs_poly: gamma_shape must be > 0");
    if (!g->keep_precision && !g->g && g->gamma_shape < 1.0)
        return fail(BINF_E_UNSUPPORTED, "gibbs_poly: generated gamma variates need gamma_shape >= 1 (got %g): supply g or sweep one at a time", g->gamma_shape);
    if (K > 16 || N > 1024 || pairwise_tree_height(N) > 3)
        return fail(BINF_E_UNSUPPORTED, "gibbs_poly: K=%lld > 16 or n_data=%lld > 1024 (or a pairwise tree deeper than 3) not covered (sweep with the per-step tier)", (long long)K, (long long)N);
    if (C == 0) return 0;
    if (!g->coefficients || !g->precision || !g->coefficients_out || !g->precision_out ||
        (N > 0 && (!g->xs || !g->ys)))
        return fail(BINF_E_ARG, "gibbs_poly: null buffer");
    if ((g->prior_means == nullptr) != (g->prior_vars == nullptr))
        return fail(BINF_E_ARG, "gibbs_poly: prior_means and prior_vars go together");
    if (g->n_adapt > 0 && !g->dt_chain)
        return fail(BINF_E_ARG, "gibbs_poly: adaption needs dt_chain");
    if (g->move == BINF_MOVE_HMC && !g->p0 && g->zig &&
        ((g->off_m + (uint64_t)(g->n - 1) * g->stride_m) >> 48))
        return fail(BINF_E_ARG, "gibbs_poly: ziggurat stream offsets must stay < 2^48");
    const int64_t waves_needed = (C + 7) / 8;
    if (waves_needed / 4 > 0x7ffffff0LL) return fail(BINF_E_UNSUPPORTED, "gibbs_poly: too many chains");
    const int64_t sb = C * K * (int64_t)sizeof(double), tb = C * (int64_t)sizeof(double);
    if ((g->coefficients_out != g->coefficients && overlap(g->coefficients_out, sb, g->coefficients, sb)) ||
        (g->precision_out != g->precision && overlap(g->precision_out, tb, g->precision, tb)) ||
        overlap(g->coefficients_out, sb, g->precision, tb) || overlap(g->precision_out, tb, g->coefficients, sb))
        return fail(BINF_E_ALIAS, "gibbs_poly: outputs may be exactly their inputs, not a partial overlap");

    PolyChainArgs a = {};
    a.theta0 = g->coefficients; a.tau0 = g->precision; a.theta_out = g->coefficients_out;
    a.tau_out = g->precision_out; a.rec_theta = g->rec_coefficients; a.rec_tau = g->rec_precision;
    a.accepted = g->accepted; a.n_accepted = g->n_accepted; a.e_before = g->e_before;
    a.e_after = g->e_after; a.xs = g->xs; a.ys = g->ys; a.prior_means = g->prior_means;
    a.prior_vars = g->prior_vars; a.p0 = g->p0; a.u = g->u; a.g = g->g; a.dt_chain = g->dt_chain;
    a.timestep = g->timestep; a.uprate = g->uprate; a.downrate = g->downrate;
    a.stepsize = g->stepsize; a.gp_shape_m1 = g->gp_shape - 1.0; a.gp_rate = g->gp_rate;
    a.g_shape = g->gamma_shape; a.g_rate = g->gamma_rate; a.C = C; a.chain_offset = g->chain_offset;
    a.seed_m = g->seed_m; a.off_m = g->off_m; a.stride_m = g->stride_m;
    a.seed_u = g->seed_u; a.off_u = g->off_u; a.stride_u = g->stride_u;
    a.seed_g = g->seed_g; a.off_g = g->off_g; a.stride_g = g->stride_g;
    a.K = (int32_t)K; a.N = (int32_t)N; a.H = pairwise_tree_height(N);
    a.tcount = poly_chain_tcount(a.N, a.H);
    a.nsteps = g->nsteps; a.n = g->n; a.thin = g->thin;
    a.n_adapt = g->move == BINF_MOVE_HMC ? g->n_adapt : 0;
    a.prior_first = g->prior_first ? 1 : 0; a.gp_where = g->gp_where; a.zig = g->zig ? 1 : 0;
    a.keep_tau = g->keep_precision ? 1 : 0;
    const bool fma = g->mode == BINF_MODE_FMA;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    if (K <= 4)      e = launch_gibbs_k<4>(a, g->move, fma, st);
    else if (K <= 8) e = launch_gibbs_k<8>(a, g->move, fma, st);
    else             e = launch_gibbs_k<16>(a, g->move, fma, st);
    if (e != hipSuccess) return hip_fail(e, "gibbs_poly launch");
    return 0;
}

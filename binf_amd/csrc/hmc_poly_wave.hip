// Fused HMC transition on the example's polynomial posterior, one lane GROUP per
// chain (n_data <= 1024, K <= 16 coefficients): the single-transition instantiation
// of poly_chain_kernel.hpp (mapping, summation orders and reference lines are
// documented there).  Same entry point as the one-lane-per-chain kernel of
// hmc_poly.hip (binf_hmc_sample_poly_f64); the per-step tier with its MFMA gradient
// takes over where chains x data is large.
#include "hmc_poly_args.hpp"
#include "poly_chain_kernel.hpp"

namespace binf {

int32_t poly_chain_tcount(int32_t N, int32_t H)
{
    int32_t longest = 0;
    for (int32_t g = 0; g < (1 << H); ++g) {
        const Leaf L = pairwise_leaf(N, H, g);
        if (L.len > longest) longest = L.len;
    }
    const int32_t tc = (longest + 7) / 8;
    return tc < 1 ? 1 : tc;
}

template <int KMAX>
static hipError_t launch_poly_wave_k(const PolyChainArgs &a, bool fma, hipStream_t st)
{
    const int64_t chains_per_wave = 64 >> (3 + a.H);
    const int64_t waves = (a.C + chains_per_wave - 1) / chains_per_wave;
    const dim3 grid((unsigned)((waves + 3) / 4));
    if (fma) poly_chain_kernel<KMAX, true, false, POLY_MOVE_HMC><<<grid, 256, 0, st>>>(a);
    else     poly_chain_kernel<KMAX, false, false, POLY_MOVE_HMC><<<grid, 256, 0, st>>>(a);
    return hipGetLastError();
}

int32_t launch_poly_wave_from(const PolyHmcArgs &h, bool fma, hipStream_t st)
{
    PolyChainArgs a = {};
    a.theta0 = h.q0; a.tau0 = h.tau_chain; a.tau = h.tau; a.theta_out = h.q_out;
    a.accepted = h.accepted; a.n_accepted = h.n_accepted; a.e_before = h.e_before;
    a.e_after = h.e_after; a.xs = h.xs; a.ys = h.ys; a.prior_means = h.prior_means;
    a.prior_vars = h.prior_vars; a.lp_pre = h.lp_pre; a.lp_post = h.lp_post; a.p0 = h.p0;
    a.u = h.u; a.dt_chain = h.dt_chain; a.timestep = h.timestep; a.uprate = h.uprate;
    a.downrate = h.downrate; a.C = h.C; a.K = h.K; a.N = h.N;
    a.H = pairwise_tree_height(h.N);
    a.tcount = poly_chain_tcount(h.N, a.H);
    a.nsteps = h.nsteps; a.n = 1; a.thin = 1; a.n_adapt = h.adapt ? 1 : 0;
    a.prior_first = h.prior_first;
    hipError_t e;
    if (a.K <= 4)      e = launch_poly_wave_k<4>(a, fma, st);
    else if (a.K <= 8) e = launch_poly_wave_k<8>(a, fma, st);
    else               e = launch_poly_wave_k<16>(a, fma, st);
    if (e != hipSuccess) return hip_fail(e, "poly_chain_kernel launch");
    return 0;
}

}  // namespace binf

// Random-walk Metropolis subsampler of the example application, chain-batched:
// RWMCSampler.sample (binf/example/samplers.py:78-92)
//
//     E_old    = -pdf.log_prob(coefficients=state)
//     change   = np.random.uniform(low=-stepsize, high=stepsize, size=len(state))
//     proposal = state + change
//     E_new    = -pdf.log_prob(coefficients=proposal)
//     accepted = np.random.random() < np.exp(-(E_new - E_old))      # PLAIN np.exp
//
// as two launches around the pdf's own evaluation of the proposal: the proposal
// kernel and the accept / select kernel.  The draws are either supplied (parity
// with the reference's np.random stream: samplers/rng.py:HostLegacyRNG) or
// generated here from the Philox stream of rng.hip, keyed by the GLOBAL element
// / chain index -- element (c, k) of the proposal draw is element
// (chain_offset + c) * K + k of binf_rng_uniform_f64(seed, offset), the
// acceptance draw of chain c is element chain_offset + c of (seed, offset) -- so
// nothing crosses PCIe and a shard of a run draws what the whole run draws.
// gfx950, wave64.
#include "gauss_common.hpp"
#include "philox_draws.hpp"

namespace binf {

__global__ void __launch_bounds__(256)
rwmc_propose_kernel(const double *state, const double *change, double *proposal, double stepsize,
                    int64_t n, int64_t e0, uint64_t seed, uint64_t offset)
{
    const double low = -stepsize;
    const double scale = stepsize - low;            // high - low, as numpy forms it
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * 256) {
        // legacy np.random.uniform: low + (high - low) * random_sample()
        const double ch = change ? change[i] : low + scale * uniform_elem(e0 + i, seed, offset);
        proposal[i] = state[i] + ch;
    }
}

struct RwmcAcceptArgs {
    const double *proposal;
    const double *state;
    const double *lp_old;
    const double *lp_new;
    const double *u;        // null: generated
    double *state_out;
    uint8_t *accepted;
    int64_t *n_accepted;
    int64_t C;
    int64_t K;
    int64_t chain_offset;
    uint64_t seed;
    uint64_t offset;
    int32_t lpc;            // threads per chain (power of two, 1..256)
};

__global__ void __launch_bounds__(256) rwmc_accept_kernel(const RwmcAcceptArgs a)
{
    const int lpc = a.lpc;
    const int sub = threadIdx.x & (lpc - 1);
    const int64_t c = (int64_t)blockIdx.x * (256 / lpc) + threadIdx.x / lpc;
    if (c >= a.C) return;
    // -(E_new - E_old) with E = -log_prob: (-a) - (-b) == b - a and -(b - a) == a - b
    // bit for bit (round-to-nearest is sign-symmetric), so the negations are not formed
    const double x = a.lp_new[c] - a.lp_old[c];
    const double uu = a.u ? a.u[c] : uniform_elem(a.chain_offset + c, a.seed, a.offset);
    const bool acc = uu < np_exp(x);
    const double *src = (acc ? a.proposal : a.state) + c * a.K;
    double *dst = a.state_out + c * a.K;
    if (dst != src)
        for (int64_t i = sub; i < a.K; i += lpc) dst[i] = src[i];
    if (sub == 0) {
        if (a.accepted) a.accepted[c] = acc ? 1 : 0;
        if (a.n_accepted && acc) a.n_accepted[c] += 1;
    }
}

}  // namespace binf

using namespace binf;

extern "C" int32_t binf_rwmc_propose_f64(const double *state, const double *change,
                                         double *proposal, double stepsize, int64_t C, int64_t K,
                                         uint64_t seed, uint64_t offset, int64_t chain_offset,
                                         void *stream)
{
    if (C < 0 || K < 0 || chain_offset < 0)
        return fail(BINF_E_ARG, "rwmc_propose: need C>=0, K>=0, chain_offset>=0");
    if (C == 0 || K == 0) return 0;
    if (!state || !proposal) return fail(BINF_E_ARG, "rwmc_propose: null buffer");
    if (C > 0x7fffffffffffffffLL / K) return fail(BINF_E_ARG, "rwmc_propose: C*K overflows");
    const int64_t n = C * K;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    rwmc_propose_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(
        state, change, proposal, stepsize, n, chain_offset * K, seed, offset);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rwmc_propose launch");
    return 0;
}

extern "C" int32_t binf_rwmc_accept_f64(const double *proposal, const double *state,
                                        const double *lp_old, const double *lp_new,
                                        const double *u, double *state_out, uint8_t *accepted,
                                        int64_t *n_accepted, int64_t C, int64_t K, uint64_t seed,
                                        uint64_t offset, int64_t chain_offset, void *stream)
{
    if (C < 0 || K < 0 || chain_offset < 0)
        return fail(BINF_E_ARG, "rwmc_accept: need C>=0, K>=0, chain_offset>=0");
    if (C == 0) return 0;
    if (!proposal || !state || !lp_old || !lp_new || !state_out)
        return fail(BINF_E_ARG, "rwmc_accept: null buffer");
    const int64_t bytes = C * K * (int64_t)sizeof(double);
    const char *o = (const char *)state_out, *p = (const char *)proposal, *s = (const char *)state;
    if ((o != p && o < p + bytes && p < o + bytes) || (o != s && o < s + bytes && s < o + bytes))
        return fail(BINF_E_ALIAS, "rwmc_accept: state_out may be exactly proposal or exactly "
                    "state, not a partial overlap");
    RwmcAcceptArgs a;
    a.proposal = proposal; a.state = state; a.lp_old = lp_old; a.lp_new = lp_new; a.u = u;
    a.state_out = state_out; a.accepted = accepted; a.n_accepted = n_accepted; a.C = C; a.K = K;
    a.chain_offset = chain_offset; a.seed = seed; a.offset = offset;
    int lpc = 1;
    while (lpc < 256 && lpc < K) lpc <<= 1;
    a.lpc = lpc;
    const int64_t blocks = (C + 256 / lpc - 1) / (256 / lpc);
    if (blocks > 0x7fffffffLL) return fail(BINF_E_UNSUPPORTED, "rwmc_accept: too many chains");
    rwmc_accept_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rwmc_accept launch");
    return 0;
}

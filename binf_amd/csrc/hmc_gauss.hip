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
// HBM sees q0, p0 once in and q_out once out.  Energy reductions are an
// in-lane sequential sum, xor-shuffles 1,2,4 inside the leaf (numpy's
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))), the leaf's tail elements, then
// xor-shuffles 8,16,32 up the leaf tree: bit-identical to np.sum.
#include "common.hpp"

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

// np.sum over one chain of the values a[t] held by the chain's lanes.
// All lanes of the wave must call this (shuffles run under a full exec mask).
template <int TMAX, bool REGULAR>
__device__ inline double chain_np_sum(const double (&a)[TMAX], int T, int rem,
                                      int lane, int H, int leafdepth)
{
    double r;
    if (REGULAR) {
        r = a[0];
#pragma unroll
        for (int t = 1; t < TMAX; ++t) r = r + a[t];
    } else {
        r = a[0];
#pragma unroll
        for (int t = 1; t < TMAX; ++t) {
            double s = r + a[t];
            r = (t < T) ? s : r;
        }
    }
    // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
    r = r + shfl_xor_f64(r, 1);
    r = r + shfl_xor_f64(r, 2);
    r = r + shfl_xor_f64(r, 4);
    double res = r;
    if (!REGULAR) {
        // n < 8: no accumulators, numpy starts from -0.0 and adds in order
        res = (T > 0) ? r : -0.0;
        // tail elements 8T .. 8T+rem-1 live at slot t == T of lanes j < rem
        double tail = 0.0;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) tail = (t == T) ? a[t] : tail;
        const int leafbase = lane & ~7;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            double v = shfl_f64(tail, leafbase + i);
            double s = res + v;
            res = (i < rem) ? s : res;
        }
    }
    // join the leaves: level l combines the two depth-(H-l) subtrees
    for (int l = 0; l < H; ++l) {
        double o = shfl_xor_f64(res, 8 << l);
        double s = res + o;
        res = (leafdepth >= H - l) ? s : res;
    }
    return 0.0 + res;   // np.add.reduce starts from the identity +0.0
}

template <bool UNIT>
__device__ inline double gauss_grad(double q, double k, double x0)
{
    // k*(x - x0), binf/pdf/__init__.py:191.  For k == 1, x0 == 0 both
    // operations are exact identities, so skipping them changes no bit.
    return UNIT ? q : k * (q - x0);
}

template <bool FMA>
__device__ inline double kick(double p, double dt, double g)
{
    return FMA ? __builtin_fma(-dt, g, p) : p - dt * g;   // hmc.py:116,120,123
}

template <bool FMA>
__device__ inline double drift(double q, double p, double dt)
{
    return FMA ? __builtin_fma(p, dt, q) : q + p * dt;    // hmc.py:119,122
}

template <int TMAX, bool REGULAR, bool UNIT, bool FMA>
__global__ void __launch_bounds__(256)
hmc_gauss_wave_kernel(const GaussArgs a)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int H = a.H;
    const int lg = 3 + H;                  // log2(lanes per chain)
    const int slot = lane & ((1 << lg) - 1);
    const int j = slot & 7;
    const int64_t chain_raw = (wave << (6 - lg)) + (lane >> lg);
    const bool cvalid = chain_raw < a.C;
    const int64_t chain = cvalid ? chain_raw : a.C - 1;

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

    const int64_t base = chain * (int64_t)a.D + off + j;
    const double *gq = a.q0 + base;
    const double *gp = a.p0 + base;

    double q[TMAX], p[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const bool m = REGULAR || (8 * t + j < n);
        q[t] = m ? gq[8 * t] : 0.0;
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const bool m = REGULAR || (8 * t + j < n);
        p[t] = m ? gp[8 * t] : 0.0;
    }

    const double dt = a.dt_chain ? a.dt_chain[chain] : a.timestep;
    const double hdt = 0.5 * dt;              // "0.5 * timestep" formed first
    const double uu = a.u[chain];
    const double c_lp = -0.5 * a.k;           // "-0.5 * k", pdf/__init__.py:185

    double sq[TMAX];
    // E_before = V(q) + 0.5*np.sum(p**2)                       hmc.py:148
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const double d = UNIT ? q[t] : q[t] - a.x0;
        sq[t] = d * d;
    }
    double Sq = chain_np_sum<TMAX, REGULAR>(sq, T, rem, lane, H, leafdepth);
#pragma unroll
    for (int t = 0; t < TMAX; ++t) sq[t] = p[t] * p[t];
    double Sp = chain_np_sum<TMAX, REGULAR>(sq, T, rem, lane, H, leafdepth);
    const double Eb = -(c_lp * Sq) + 0.5 * Sp;

    // _leapfrog                                                hmc.py:116-123
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
        p[t] = kick<FMA>(p[t], hdt, gauss_grad<UNIT>(q[t], a.k, a.x0));
    for (int s = 0; s < a.nsteps - 1; ++s) {
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            q[t] = drift<FMA>(q[t], p[t], dt);
            p[t] = kick<FMA>(p[t], dt, gauss_grad<UNIT>(q[t], a.k, a.x0));
        }
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        q[t] = drift<FMA>(q[t], p[t], dt);
        p[t] = kick<FMA>(p[t], hdt, gauss_grad<UNIT>(q[t], a.k, a.x0));
    }

    // E_after                                                  hmc.py:150
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const double d = UNIT ? q[t] : q[t] - a.x0;
        sq[t] = d * d;
    }
    Sq = chain_np_sum<TMAX, REGULAR>(sq, T, rem, lane, H, leafdepth);
#pragma unroll
    for (int t = 0; t < TMAX; ++t) sq[t] = p[t] * p[t];
    Sp = chain_np_sum<TMAX, REGULAR>(sq, T, rem, lane, H, leafdepth);
    const double Ea = -(c_lp * Sq) + 0.5 * Sp;

    // acc = uniform < exp(-(E_after - E_before)), csb clipped exp  hmc.py:151
    double x = -(Ea - Eb);
    x = (x < -308.0) ? -308.0 : x;
    x = (x > 709.0) ? 709.0 : x;          // NaN falls through both, as np.clip
    const bool acc = uu < exp(x);

    if (cvalid && slot == 0) {
        a.accepted[chain] = acc ? 1 : 0;
        if (a.n_accepted && acc) a.n_accepted[chain] += 1;   // hmc.py:161
        if (a.e_before) a.e_before[chain] = Eb;
        if (a.e_after) a.e_after[chain] = Ea;
        if (a.adapt)                                         // hmc.py:188-191
            a.dt_chain[chain] = acc ? dt * a.uprate : dt * a.downrate;
    }

    // return value: the proposal if accepted, else the old state  hmc.py:159-164
    double *go = a.q_out + base;
    if (cvalid && canonical) {
        if (acc) {
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (REGULAR || (8 * t + j < n)) go[8 * t] = q[t];
        } else if (a.q_out != a.q0) {
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (REGULAR || (8 * t + j < n)) go[8 * t] = gq[8 * t];
        }
    }
}

template <int TMAX, bool REGULAR>
static hipError_t launch_tr(const GaussArgs &a, bool unit, bool fma,
                            dim3 grid, hipStream_t st)
{
    if (unit) {
        if (fma) hmc_gauss_wave_kernel<TMAX, REGULAR, true, true><<<grid, 256, 0, st>>>(a);
        else     hmc_gauss_wave_kernel<TMAX, REGULAR, true, false><<<grid, 256, 0, st>>>(a);
    } else {
        if (fma) hmc_gauss_wave_kernel<TMAX, REGULAR, false, true><<<grid, 256, 0, st>>>(a);
        else     hmc_gauss_wave_kernel<TMAX, REGULAR, false, false><<<grid, 256, 0, st>>>(a);
    }
    return hipGetLastError();
}

template <int TMAX>
static hipError_t launch_t(const GaussArgs &a, bool regular, bool unit, bool fma,
                           dim3 grid, hipStream_t st)
{
    return regular ? launch_tr<TMAX, true>(a, unit, fma, grid, st)
                   : launch_tr<TMAX, false>(a, unit, fma, grid, st);
}

}  // namespace binf

using namespace binf;

extern "C" int32_t binf_hmc_sample_gauss_f64(
    const double *q0, const double *p0, const double *u, double *q_out,
    uint8_t *accepted, int64_t *n_accepted, double *e_before, double *e_after,
    double timestep,
    double *dt_chain, int64_t C, int64_t D, int32_t nsteps, double k, double x0,
    int32_t adapt, double uprate, double downrate, int32_t mode, void *stream)
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
    a.q0 = q0; a.p0 = p0; a.u = u; a.q_out = q_out; a.accepted = accepted; a.n_accepted = n_accepted;
    a.e_before = e_before; a.e_after = e_after; a.dt_chain = dt_chain;
    a.timestep = timestep; a.k = k; a.x0 = x0; a.uprate = uprate;
    a.downrate = downrate; a.C = C; a.D = (int32_t)D; a.nsteps = nsteps;
    a.H = H; a.adapt = adapt;

    const int64_t chains_per_wave = 64 >> (3 + H);
    const int64_t waves = (C + chains_per_wave - 1) / chains_per_wave;
    const int64_t blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffffLL)
        return fail(BINF_E_UNSUPPORTED, "hmc_sample_gauss: too many chains");
    dim3 grid((unsigned)blocks);
    hipStream_t st = (hipStream_t)stream;
    const bool unit = (k == 1.0 && x0 == 0.0);
    const bool fma = (mode == BINF_MODE_FMA);
    hipError_t e;
    if (tneed <= 1)       e = launch_t<1>(a, regular && tneed == 1, unit, fma, grid, st);
    else if (tneed <= 2)  e = launch_t<2>(a, regular && tneed == 2, unit, fma, grid, st);
    else if (tneed <= 4)  e = launch_t<4>(a, regular && tneed == 4, unit, fma, grid, st);
    else if (tneed <= 8)  e = launch_t<8>(a, regular && tneed == 8, unit, fma, grid, st);
    else if (tneed <= 12) e = launch_t<12>(a, regular && tneed == 12, unit, fma, grid, st);
    else                  e = launch_t<16>(a, regular && tneed == 16, unit, fma, grid, st);
    if (e != hipSuccess) return hip_fail(e, "hmc_gauss_wave_kernel launch");
    return 0;
}

// Device helpers shared by the fused Gaussian HMC kernels (single-transition
// and persistent multi-transition).  gfx950, wave64.
#pragma once
#include "common.hpp"

namespace binf {

// Arguments of the fused Gaussian HMC kernels (hmc_gauss.hip, hmc_gauss_split.hip)
struct GaussNArgs {
    const double *q0;
    const double *p0;        // [n x C x D]
    const double *u;         // [n x C]
    double *q_out;           // [C x D]
    double *samples;         // [n/thin x C x D] or null
    uint8_t *accepted;       // [n x C] or null
    int64_t *n_accepted;     // [C] or null
    double *e_before;        // [n x C] or null
    double *e_after;         // [n x C] or null
    double *dt_chain;        // [C] or null
    double timestep;
    double k;
    double x0;
    double uprate;
    double downrate;
    int64_t C;
    int32_t D;
    int32_t nsteps;
    int32_t H;
    int32_t n;               // transitions per launch
    int32_t thin;            // record every thin-th state (>= 1)
    int32_t n_adapt;         // the first n_adapt transitions adapt the timestep
    int32_t stagger;         // start delay per SIMD wave slot, units of ~64 cycles (0 = none)
    int32_t force_lds_stash; // development aid (BINF_GAUSS_STASH=lds): always stash the state in LDS
    // draws generated in the kernel (hmc_gauss_rng.hip)
    uint64_t rng_seed;
    uint64_t rng_offset;
    int64_t chain_offset;    // global index of this launch's first chain (sharded runs)
    double *p_dump;          // [n x C x D], GAUSS_RNG_DUMP only
    double *u_dump;          // [n x C],     GAUSS_RNG_DUMP only
};

// How a [C x D] batch maps onto waves (host side; shared by the launchers)
struct GaussPlan {
    int32_t H;        // height of numpy's pairwise tree for length D
    int LW;           // log2(waves per chain)
    int tneed;        // elements per lane
    bool regular;     // all leaves equal, full depth, multiples of 8
    int64_t blocks;   // workgroups of the one-wave-per-chain / wide kernels
};

inline GaussPlan gauss_plan(int64_t C, int64_t D)
{
    GaussPlan p;
    p.H = pairwise_tree_height(D);
    p.LW = p.H > 3 ? p.H - 3 : 0;
    p.tneed = 1;
    p.regular = true;
    int32_t len0 = -1;
    for (int g = 0; g < (1 << p.H); ++g) {
        Leaf L = pairwise_leaf((int32_t)D, p.H, g);
        int tn = (L.len + 7) / 8;
        if (tn > p.tneed) p.tneed = tn;
        if (len0 < 0) len0 = L.len;
        if (L.len != len0 || L.depth != p.H || (L.len & 7)) p.regular = false;
    }
    if (p.LW == 0) {
        const int64_t chains_per_wave = 64 >> (3 + p.H);
        const int64_t waves = (C + chains_per_wave - 1) / chains_per_wave;
        p.blocks = (waves + 3) / 4;
    } else {
        const int64_t chains_per_block = (p.LW == 3 ? 8 : 4) >> p.LW;
        p.blocks = (C + chains_per_block - 1) / chains_per_block;
    }
    return p;
}

// hmc_gauss_split.hip: few chains of D in {768, 1024} spread over 2 / 4 waves each
int gauss_split_factor(int64_t C, int32_t H, bool regular, int tneed);
int gauss_stagger(int n);
int gauss_force_lds_stash();
hipError_t launch_gauss_split(const GaussNArgs &a, int tneed, int split, bool unit, bool fma,
                              hipStream_t st);

// In-lane part of np.sum: numpy's j-th accumulator of a leaf adds a[8t+j] for
// t = 0..T-1 in order; element t == T (if the lane has one) is a tail element
// and is added after the leaf's accumulators have been combined.
struct LaneSum {
    double r;
    double tail;
};

template <bool REGULAR>
__device__ inline void lane_sum_add(LaneSum &s, double v, int t, int T)
{
    if (REGULAR) {
        s.r = (t == 0) ? v : s.r + v;
    } else {
        const double n = s.r + v;
        s.r = (t == 0) ? v : ((t < T) ? n : s.r);
        s.tail = (t == T) ? v : s.tail;
    }
}

// Cross-lane part of np.sum.  All lanes of the wave must call this (the
// shuffles need a full exec mask).  LW = log2(waves per chain): levels 0..2 of
// the leaf tree live inside a wave (xor-shuffles 8, 16, 32), levels >= 3 join
// the waves of a chain through the LDS slots `xch` (one per wave of the
// block; every wave of the block must call this the same number of times).
template <bool REGULAR, int LW = 0>
__device__ inline double chain_sum_finish(const LaneSum &s, int T, int rem,
                                          int lane, int H, int leafdepth,
                                          double *xch = nullptr, int wib = 0)
{
    // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
    const double r = sum8_f64(s.r);
    double res = r;
    if (!REGULAR) {
        // n < 8: no accumulators, numpy starts from -0.0 and adds in order
        res = (T > 0) ? r : -0.0;
        const int leafbase = lane & ~7;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const double v = shfl_f64(s.tail, leafbase + i);
            const double n = res + v;
            res = (i < rem) ? n : res;
        }
    }
    // join the leaves: level l combines the two depth-(H-l) subtrees
    const int hin = (LW > 0) ? 3 : H;
    for (int l = 0; l < hin; ++l) {
        const double o = xor_level_f64(res, l, lane);
        const double n = res + o;
        res = (leafdepth >= H - l) ? n : res;
    }
    if (LW > 0) {
#pragma unroll
        for (int l = 3; l < 3 + LW; ++l) {
            __syncthreads();
            if (lane == 0) xch[wib] = res;
            __syncthreads();
            const double o = xch[wib ^ (1 << (l - 3))];
            const double n = res + o;
            res = (leafdepth >= H - l) ? n : res;
        }
    }
    return 0.0 + res;   // np.add.reduce starts from the identity +0.0
}

// A double constant held in an SGPR pair at the point of use.  The volatile
// asm keeps LLVM from hoisting it out of the transition loop into a VGPR pair
// (the hoisted exp() constants alone cost the persistent kernel 20 VGPRs and
// with them the fourth wave per SIMD).
__device__ inline double sgpr_const(unsigned long long bits)
{
    double c = __longlong_as_double((long long)bits);
    asm volatile("" : "+s"(c));
    return c;
}

// exp(x) for x in [-308, 709] (the clipped exponent of the accept test,
// hmc.py:151 + csb.numeric.exp).  Same algorithm and coefficients as the
// device math library's exp (argument reduction by ln2 in two parts, degree-11
// polynomial, ldexp), so the accept decisions are those of the library call it
// replaces; written out so that its constants stay in SGPRs.
__device__ inline double exp_clipped_range(double x)
{
    const double n = __builtin_rint(x * sgpr_const(0x3ff71547652b82feULL));     // 1/ln2
    double r = __builtin_fma(sgpr_const(0xbfe62e42fefa39efULL), n, x);          // -ln2 (hi)
    r = __builtin_fma(sgpr_const(0xbc7abc9e3b39803fULL), n, r);                 // -ln2 (lo)
    double p = __builtin_fma(sgpr_const(0x3e5ade156a5dcb37ULL), r,
                             sgpr_const(0x3e928af3fca7ab0cULL));
    p = __builtin_fma(r, p, sgpr_const(0x3ec71dee623fde64ULL));
    p = __builtin_fma(r, p, sgpr_const(0x3efa01997c89e6b0ULL));
    p = __builtin_fma(r, p, sgpr_const(0x3f2a01a014761f6eULL));
    p = __builtin_fma(r, p, sgpr_const(0x3f56c16c1852b7b0ULL));
    p = __builtin_fma(r, p, sgpr_const(0x3f81111111122322ULL));
    p = __builtin_fma(r, p, sgpr_const(0x3fa55555555502a1ULL));
    p = __builtin_fma(r, p, sgpr_const(0x3fc5555555555511ULL));
    p = __builtin_fma(r, p, sgpr_const(0x3fe000000000000bULL));
    p = __builtin_fma(r, p, 1.0);
    p = __builtin_fma(r, p, 1.0);
    return __builtin_ldexp(p, (int)n);
}

// np.exp on the whole real line (the accept test of the RWMC sampler,
// binf/example/samplers.py:86, uses numpy's exp, not csb's clipped one): +inf above
// the overflow threshold, subnormals and 0 below, NaN for NaN -- the same
// polynomial as every other accept test here.
__device__ inline double np_exp(double x)
{
    if (x > 709.782712893384) return __builtin_inf();
    if (x < -745.2) return 0.0;
    return exp_clipped_range(x);
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

}  // namespace binf

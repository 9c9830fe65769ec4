// The random draws of HMCSampler.sample() -- np.random.normal(size=q.shape) and
// np.random.uniform() (binf/samplers/hmc.py:146,151) -- generated INSIDE the
// sampling kernel: one xoshiro128++ stream (Blackman & Vigna 2018) per lane,
// normals by a 1024-layer ziggurat (Marsaglia & Tsang 2000, Doornik's ZIGNOR
// acceptance tests) whose layer table lives in 8 KiB of LDS.
//
// Why not the Philox generator of rng.hip: Philox4x32-10 costs ~100 integer
// instructions per two normals, four of them quarter-rate 32-bit multiplies per
// round; xoshiro128++ is 10 full-rate add / xor / rotate instructions per 32 bits.
// The stream of lane `id` for launch (seed, offset) starts from the Philox block
// (id, offset) under key seed ^ domain tag, so streams are reproducible, do not
// depend on which wave or workgroup runs the lane, and never share an offset with
// the stand-alone kernels.  NOT stream-compatible with numpy's MT19937: parity
// runs inject host draws (samplers/rng.py:HostLegacyRNG).
#pragma once
#include "gauss_common.hpp"
#include "philox.hpp"
#include "zig_tables.hpp"

namespace binf {

// the ziggurat of the fused generator: 1024 layers, table = ZIG_X (8 KiB of LDS)
constexpr int XZIG_C = ZIG_C;
constexpr int XZIG_BITS = ZIG_BITS;
constexpr double XZIG_TAIL_R = ZIG_TAIL_R;
#define XZIG_TABLE ZIG_X

struct Xo128 {
    uint32_t s0, s1, s2, s3;
};

__device__ inline uint32_t rotl32(uint32_t x, int k)
{
    return (x << k) | (x >> (32 - k));
}

__device__ inline uint32_t xo_next(Xo128 &g)
{
    const uint32_t r = rotl32(g.s0 + g.s3, 7) + g.s0;
    const uint32_t t = g.s1 << 9;
    g.s2 ^= g.s0;
    g.s3 ^= g.s1;
    g.s1 ^= g.s2;
    g.s0 ^= g.s3;
    g.s2 ^= t;
    g.s3 = rotl32(g.s3, 11);
    return r;
}

__device__ inline Xo128 xo_seed(uint64_t stream, uint64_t seed, uint64_t offset)
{
    const Philox4 r = philox4x32_10((uint32_t)stream, (uint32_t)(stream >> 32), (uint32_t)offset,
                                    (uint32_t)(offset >> 32), (uint32_t)seed,
                                    (uint32_t)(seed >> 32) ^ 0x58534f52u);   // domain tag
    Xo128 g = {r.v[0], r.v[1], r.v[2], r.v[3]};
    if ((g.s0 | g.s1 | g.s2 | g.s3) == 0) g.s0 = 1;            // the one forbidden state
    return g;
}

// 53-bit uniform in [0, 1) from two outputs (numpy's random_sample recipe)
__device__ inline double xo_uniform53(Xo128 &g)
{
    const uint32_t a = xo_next(g), b = xo_next(g);
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// One ziggurat candidate from two outputs: layer = top 10 bits of the first,
// u in [-1, 1) from the other 52 bits (built as a double in [1, 2), then
// 2 d - 3 exactly).  x = u * X[layer]; the candidate is final iff |x| < X[layer+1]
// (inside the part of the layer that lies wholly under the density).
__device__ inline double xzig_candidate(Xo128 &g, const double *zx, int &layer, bool &ok)
{
    const uint32_t hi = xo_next(g), lo = xo_next(g);
    layer = (int)(hi >> (32 - XZIG_BITS));
    const double d = __hiloint2double((int)(0x3ff00000u | (hi & 0xfffffu)), (int)lo);
    const double u = __builtin_fma(2.0, d, -3.0);
    const double x = u * zx[layer];
    ok = __builtin_fabs(x) < zx[layer + 1];
    return x;
}

__device__ inline double xzig_tail(Xo128 &g, bool neg)
{
    double x = 0.0;
    for (int t = 0; t < 64; ++t) {
        x = log(1.0 - xo_uniform53(g)) / XZIG_TAIL_R;         // <= 0
        const double y = log(1.0 - xo_uniform53(g));
        if (-2.0 * y >= x * x) break;
    }
    return neg ? x - XZIG_TAIL_R : XZIG_TAIL_R - x;
}

// Finish a candidate that failed the fast test: wedge test of its layer (or the
// tail for the base layer); on rejection draw fresh candidates until one is
// accepted.
__device__ inline double xzig_resolve(double x, int layer, Xo128 &g, const double *zx)
{
    for (int k = 0; k < 64; ++k) {
        if (layer == 0) return xzig_tail(g, x < 0.0);
        const double x2 = x * x;
        const double f0 = exp_clipped_range(-0.5 * (zx[layer] * zx[layer] - x2));
        const double f1 = exp_clipped_range(-0.5 * (zx[layer + 1] * zx[layer + 1] - x2));
        if (f1 + xo_uniform53(g) * (f0 - f1) < 1.0) return x;
        bool ok;
        x = xzig_candidate(g, zx, layer, ok);
        if (ok) return x;
    }
    return x;
}

// N normals into out[0..N-1]; `want` bit i clear = element i is not drawn (lanes
// past the end of a ragged leaf).  First the N candidates, then this lane's
// rejections in index order: a wave walks the slow path max-over-lanes times
// per N * 64 candidates instead of once per failing candidate.
template <int N>
__device__ inline void xzig_normals(double (&out)[N], unsigned want, Xo128 &g, const double *zx)
{
    static_assert(N <= 9 && XZIG_BITS <= 10, "10-bit layers are packed three to a register");
    unsigned fail = 0;
    uint32_t lay[3] = {0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < N; ++i) {
        out[i] = 0.0;
        if (want & (1u << i)) {
            int layer;
            bool ok;
            out[i] = xzig_candidate(g, zx, layer, ok);
            if (!ok) fail |= 1u << i;
            lay[i / 3] |= (uint32_t)layer << (10 * (i % 3));
        }
    }
    while (fail) {
        const int i = __ffs(fail) - 1;
        fail &= fail - 1;
        uint32_t word = 0;
        double x = 0.0;
#pragma unroll
        for (int e = 0; e < N; ++e) {
            x = (e == i) ? out[e] : x;
            word = (e == i) ? (lay[e / 3] >> (10 * (e % 3))) : word;
        }
        const double v = xzig_resolve(x, (int)(word & 0x3ffu), g, zx);
#pragma unroll
        for (int e = 0; e < N; ++e) out[e] = (e == i) ? v : out[e];
    }
}

// LDS copy of the layer table (all threads of the block; a barrier follows)
__device__ inline void xzig_load_table(double *zx, int tid, int nthreads)
{
    for (int k = tid; k <= XZIG_C; k += nthreads) zx[k] = XZIG_TABLE[k];
}

}  // namespace binf

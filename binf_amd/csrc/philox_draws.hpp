// Device functions of the stand-alone Philox draw streams (rng.hip): shared by the
// fill kernels there and by kernels that generate the same draws in place
// (rwmc.hip, gibbs_poly.hip).  Element i of a stream is a function of (seed,
// offset, GLOBAL index i) only.
#pragma once
#include "common.hpp"
#include "philox.hpp"
#include "zig_tables.hpp"

namespace binf {

// 53-bit uniform in [0, 1) from two 32-bit words (numpy's random_sample recipe)
__host__ __device__ inline double u53(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// two uniforms per Philox call: element pair index i -> counter (lo, hi, offset lo, offset hi)
__device__ inline void uniforms2(int64_t i, uint64_t seed, uint64_t offset, double &a, double &b)
{
    const Philox4 r = philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32),
                                    (uint32_t)offset, (uint32_t)(offset >> 32),
                                    (uint32_t)seed, (uint32_t)(seed >> 32));
    a = u53(r.v[0], r.v[1]);
    b = u53(r.v[2], r.v[3]);
}

// Box-Muller: two normals from two uniforms
__device__ inline void normals2(int64_t i, uint64_t seed, uint64_t offset, double &a, double &b)
{
    double u1, u2;
    uniforms2(i, seed, offset, u1, u2);
    const double r = sqrt(-2.0 * log(1.0 - u1));       // 1-u1 in (0, 1]
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    a = r * c;
    b = r * s;
}

// ---------------------------------------------------------------------------
// Ziggurat normals (Marsaglia & Tsang 2000, Doornik's ZIGNOR formulation, 1024
// layers, tables in zig_tables.hpp): 99.57 % of the candidates cost one 64-bit
// Philox word, a table look-up, a compare and a multiply; the Box-Muller kernel
// above spends an FP64 log, sqrt and sincospi on every pair.
//
// A rejected candidate needs two exp() and another Philox block, and on a
// 64-lane wave ONE rejecting lane makes the whole wave walk that path.  So each
// lane first tests 8 candidates (4 Philox blocks) and only then resolves its
// rejections in a short loop: the wave pays for max-over-lanes rejections per
// 512 candidates instead of per 128.
//
// Determinism: outputs 2i, 2i+1 come from block (i, offset) whatever the launch
// geometry; retries use blocks tagged (attempt, which) in the top 16 bits of
// the stream offset (so offsets must stay < 2^48).
// ---------------------------------------------------------------------------
__device__ inline Philox4 zig_block(int64_t i, uint64_t seed, uint64_t offset, uint32_t tag)
{
    return philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)offset,
                         ((uint32_t)(offset >> 32) & 0xffffu) | (tag << 16),
                         (uint32_t)seed, (uint32_t)(seed >> 32));
}

// layer index from the low ZIG_BITS bits, u in [-1, 1) from the 53 bits above
__device__ inline void zig_split(uint32_t lo, uint32_t hi, int &layer, double &u)
{
    layer = lo & (ZIG_C - 1);
    const double uu = ((double)(hi >> 1) * 4194304.0 + (double)(lo >> ZIG_BITS)) *
                      (1.0 / 9007199254740992.0);
    u = 2.0 * uu - 1.0;
}

__device__ inline double zig_tail(int64_t i, uint64_t seed, uint64_t offset, uint32_t which,
                                  bool neg)
{
    double x = 0.0;
    for (uint32_t t = 0; t < 64; ++t) {
        const Philox4 r = zig_block(i, seed, offset, 0x8000u | (t << 1) | which);
        x = log(1.0 - u53(r.v[0], r.v[1])) / ZIG_TAIL_R;      // <= 0
        const double y = log(1.0 - u53(r.v[2], r.v[3]));
        if (-2.0 * y >= x * x) break;
    }
    return neg ? x - ZIG_TAIL_R : ZIG_TAIL_R - x;
}

// resolve a candidate that failed the fast test (zx / zr: LDS copies of the tables)
__device__ inline double zig_slow(uint32_t lo, uint32_t hi, const double *zx, const double *zr,
                                  int64_t i, uint64_t seed, uint64_t offset, uint32_t which)
{
    for (uint32_t k = 1;; ++k) {
        int layer;
        double u;
        zig_split(lo, hi, layer, u);
        if (fabs(u) < zr[layer]) return u * zx[layer];
        if (layer == 0) return zig_tail(i, seed, offset, which, u < 0.0);
        const Philox4 r = zig_block(i, seed, offset, (k << 1) | which);
        const double x = u * zx[layer];
        const double x2 = x * x;
        const double f0 = exp(-0.5 * (zx[layer] * zx[layer] - x2));
        const double f1 = exp(-0.5 * (zx[layer + 1] * zx[layer + 1] - x2));
        if (f1 + u53(r.v[2], r.v[3]) * (f0 - f1) < 1.0 || k >= 63) return x;
        lo = r.v[0];
        hi = r.v[1];
    }
}

// Gamma(shape, 1), Marsaglia & Tsang (2000); shape < 1 via Gamma(shape+1)*U^(1/shape).
// Attempt k of global element i uses counter i under offset + 2k, 2k + 1 (bounded retries).
// SMALL = false drops the shape < 1 branch (and with it pow(), ~100 VGPRs when inlined
// into a larger kernel): same values for every shape >= 1.
template <bool SMALL = true>
__device__ inline double gamma_elem(int64_t i, double shape, uint64_t seed, uint64_t offset)
{
    const double alpha = shape < 1.0 ? shape + 1.0 : shape;
    const double d = alpha - 1.0 / 3.0;
    const double c = 1.0 / sqrt(9.0 * d);
    double g = d;                                   // fallback after 64 rejections
    for (int k = 0; k < 64; ++k) {
        double x, unused, u1, u2;
        normals2(i, seed, offset + 2 * (uint64_t)k, x, unused);
        uniforms2(i, seed, offset + 2 * (uint64_t)k + 1, u1, u2);
        const double t = 1.0 + c * x;
        if (t <= 0.0) continue;
        const double v = t * t * t;
        const double uu = 1.0 - u1;                 // (0, 1]
        if (log(uu) < 0.5 * x * x + d - d * v + d * log(v)) {
            g = d * v;
            if (SMALL && shape < 1.0) g *= pow(1.0 - u2, 1.0 / shape);
            break;
        }
    }
    return g;
}

// ---- single elements of the streams (for kernels that draw for ONE chain per
// lane: rwmc.hip, gibbs_poly.hip).  Same values the fill kernels write. -----------

// element e of the uniform stream (seed, offset): the block of pair e / 2 gives
// elements 2g and 2g + 1
__device__ inline double uniform_elem(int64_t e, uint64_t seed, uint64_t offset)
{
    double a, b;
    uniforms2(e >> 1, seed, offset, a, b);
    return (e & 1) ? b : a;
}

// element e of the ziggurat normal stream (seed, offset); zx / zr: LDS copies of
// ZIG_X (ZIG_C + 1 entries) and ZIG_RATIO (ZIG_C entries)
__device__ inline double zig_normal_elem(int64_t e, uint64_t seed, uint64_t offset,
                                         const double *zx, const double *zr)
{
    const int64_t i = e >> 1;
    const uint32_t which = (uint32_t)(e & 1);
    const Philox4 r = zig_block(i, seed, offset, 0);
    const uint32_t lo = which ? r.v[2] : r.v[0], hi = which ? r.v[3] : r.v[1];
    int layer;
    double u;
    zig_split(lo, hi, layer, u);
    if (fabs(u) < zr[layer]) return u * zx[layer];
    return zig_slow(lo, hi, zx, zr, i, seed, offset, which);
}

}  // namespace binf

// Device random draws for throughput mode: the momentum draw
// np.random.normal(size=q.shape) and the acceptance draw np.random.uniform()
// of HMCSampler.sample (binf/samplers/hmc.py:146,151) and the
// np.random.gamma(shape) of GammaSampler.sample (binf/example/samplers.py:47),
// generated in HBM so that nothing crosses PCIe.  gfx950, wave64.
//
// Counter-based Philox4x32-10 (Salmon et al., "Parallel random numbers: as
// easy as 1, 2, 3", SC'11): element i of a call uses counter (i, stream
// offset) under key = seed, so results do not depend on the launch geometry
// and consecutive calls never overlap.  NOT stream-compatible with numpy's
// MT19937 -- parity runs take their draws from the host (samplers/rng.py).
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

__global__ void __launch_bounds__(256)
rng_uniform_kernel(double *out, int64_t n, uint64_t seed, uint64_t offset)
{
    const int64_t np = (n + 1) / 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < np;
         i += (int64_t)gridDim.x * 256) {
        double a, b;
        uniforms2(i, seed, offset, a, b);
        out[2 * i] = a;
        if (2 * i + 1 < n) out[2 * i + 1] = b;
    }
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

__global__ void __launch_bounds__(256)
rng_normal_kernel(double *out, int64_t n, uint64_t seed, uint64_t offset)
{
    const int64_t np = (n + 1) / 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < np;
         i += (int64_t)gridDim.x * 256) {
        double a, b;
        normals2(i, seed, offset, a, b);
        out[2 * i] = a;
        if (2 * i + 1 < n) out[2 * i + 1] = b;
    }
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

__global__ void __launch_bounds__(256)
rng_normal_zig_kernel(double *out, int64_t n, uint64_t seed, uint64_t offset)
{
    __shared__ double zx[ZIG_C + 1];
    __shared__ double zr[ZIG_C];
    for (int k = threadIdx.x; k <= ZIG_C; k += 256) zx[k] = ZIG_X[k];
    for (int k = threadIdx.x; k < ZIG_C; k += 256) zr[k] = ZIG_RATIO[k];
    __syncthreads();
    const int64_t np = (n + 1) / 2;                 // Philox blocks = output pairs
    // a workgroup covers 1024 consecutive pairs per round: lane t owns pairs
    // base + t, base + 256 + t, ... so that stores are 16 B per lane, coalesced
    for (int64_t base = (int64_t)blockIdx.x * 1024; base < np;
         base += (int64_t)gridDim.x * 1024) {
        uint32_t lo[8], hi[8];
        double val[8];
        unsigned mask = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const Philox4 r = zig_block(base + 256 * b + threadIdx.x, seed, offset, 0);
            lo[2 * b] = r.v[0]; hi[2 * b] = r.v[1];
            lo[2 * b + 1] = r.v[2]; hi[2 * b + 1] = r.v[3];
        }
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            int layer;
            double u;
            zig_split(lo[d], hi[d], layer, u);
            val[d] = u * zx[layer];
            if (!(fabs(u) < zr[layer])) mask |= 1u << d;
        }
        while (mask) {
            const int d = __ffs(mask) - 1;
            mask &= mask - 1;
            uint32_t l = 0, h = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                l = (e == d) ? lo[e] : l;
                h = (e == d) ? hi[e] : h;
            }
            const double v = zig_slow(l, h, zx, zr, base + 256 * (d >> 1) + threadIdx.x, seed,
                                      offset, d & 1);
#pragma unroll
            for (int e = 0; e < 8; ++e) val[e] = (e == d) ? v : val[e];
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int64_t i = base + 256 * b + threadIdx.x;
            if (2 * i + 1 < n) {
                typedef double v2d __attribute__((ext_vector_type(2)));
                v2d w;
                w.x = val[2 * b];
                w.y = val[2 * b + 1];
                *reinterpret_cast<v2d *>(out + 2 * i) = w;
            } else if (2 * i < n) {
                out[2 * i] = val[2 * b];
            }
        }
    }
}

// Gamma(shape, 1), Marsaglia & Tsang (2000); shape < 1 via Gamma(shape+1)*U^(1/shape).
// Attempt k of element i uses counter i under offset + k (bounded retries).
__global__ void __launch_bounds__(256)
rng_gamma_kernel(double *out, int64_t n, double shape, uint64_t seed, uint64_t offset)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * 256) {
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
                if (shape < 1.0) g *= pow(1.0 - u2, 1.0 / shape);
                break;
            }
        }
        out[i] = g;
    }
}

static unsigned rng_grid(int64_t work)
{
    int64_t b = (work + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace binf

using namespace binf;

extern "C" int32_t binf_rng_philox4x32_10(const uint32_t counter[4], const uint32_t key[2],
                                          uint32_t out[4])
{
    if (!counter || !key || !out) return fail(BINF_E_ARG, "rng_philox: null pointer");
    const Philox4 r = philox4x32_10(counter[0], counter[1], counter[2], counter[3], key[0], key[1]);
    for (int i = 0; i < 4; ++i) out[i] = r.v[i];
    return 0;
}

extern "C" int32_t binf_rng_uniform_f64(double *out, int64_t n, uint64_t seed,
                                        uint64_t offset, void *stream)
{
    if (n < 0) return fail(BINF_E_ARG, "rng_uniform: negative size");
    if (n == 0) return 0;
    if (!out) return fail(BINF_E_ARG, "rng_uniform: null buffer");
    rng_uniform_kernel<<<dim3(rng_grid((n + 1) / 2)), 256, 0, (hipStream_t)stream>>>(out, n, seed, offset);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rng_uniform launch");
    return 0;
}

extern "C" int32_t binf_rng_normal_f64(double *out, int64_t n, uint64_t seed,
                                       uint64_t offset, void *stream)
{
    if (n < 0) return fail(BINF_E_ARG, "rng_normal: negative size");
    if (n == 0) return 0;
    if (!out) return fail(BINF_E_ARG, "rng_normal: null buffer");
    rng_normal_kernel<<<dim3(rng_grid((n + 1) / 2)), 256, 0, (hipStream_t)stream>>>(out, n, seed, offset);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rng_normal launch");
    return 0;
}

extern "C" int32_t binf_rng_gamma_f64(double *out, int64_t n, double shape, uint64_t seed,
                                      uint64_t offset, void *stream)
{
    if (n < 0 || !(shape > 0.0)) return fail(BINF_E_ARG, "rng_gamma: need n >= 0 and shape > 0");
    if (n == 0) return 0;
    if (!out) return fail(BINF_E_ARG, "rng_gamma: null buffer");
    rng_gamma_kernel<<<dim3(rng_grid(n)), 256, 0, (hipStream_t)stream>>>(out, n, shape, seed, offset);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rng_gamma launch");
    return 0;
}

extern "C" int32_t binf_rng_normal_zig_f64(double *out, int64_t n, uint64_t seed,
                                           uint64_t offset, void *stream)
{
    if (n < 0) return fail(BINF_E_ARG, "rng_normal_zig: negative size");
    if (offset >> 48) return fail(BINF_E_ARG, "rng_normal_zig: offset must be < 2^48");
    if (n == 0) return 0;
    if (!out) return fail(BINF_E_ARG, "rng_normal_zig: null buffer");
    rng_normal_zig_kernel<<<dim3(rng_grid((n + 7) / 8)), 256, 0, (hipStream_t)stream>>>(out, n, seed, offset);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rng_normal_zig launch");
    return 0;
}

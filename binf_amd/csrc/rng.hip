// Device random draws for throughput mode: the momentum draw
// np.random.normal(size=q.shape) and the acceptance draw np.random.uniform()
// of HMCSampler.sample (binf/samplers/hmc.py:146,151) and the
// np.random.gamma(shape) of GammaSampler.sample (binf/example/samplers.py:47),
// generated in HBM so that nothing crosses PCIe.  gfx950, wave64.
//
// Counter-based Philox4x32-10 (Salmon et al., "Parallel random numbers: as
// easy as 1, 2, 3", SC'11): element i of a call uses counter (i, stream
// offset) under key = seed, so results do not depend on the launch geometry
// and consecutive calls never overlap.  `i` is the GLOBAL element index: a
// call that fills a window of a larger logical array (one rank's shard of the
// chains) passes the window's first index as elem_offset and gets exactly the
// values the unsharded call writes there.  NOT stream-compatible with numpy's
// MT19937 -- parity runs take their draws from the host (samplers/rng.py).
#include "philox_draws.hpp"

namespace binf {

// Pairs of outputs come from one Philox block: global elements 2g, 2g+1 from
// block g.  A window [e0, e0 + n) touches blocks g0 = e0/2 .. (e0+n-1)/2.
__device__ inline void store_pair(double *out, int64_t n, int64_t e0, int64_t g, double a, double b)
{
    const int64_t l = 2 * g - e0;            // local index of the pair's first element
    if (l >= 0 && l < n) out[l] = a;
    if (l + 1 >= 0 && l + 1 < n) out[l + 1] = b;
}

__global__ void __launch_bounds__(256)
rng_uniform_kernel(double *out, int64_t n, uint64_t seed, uint64_t offset, int64_t e0)
{
    const int64_t g0 = e0 >> 1;
    const int64_t np = ((e0 + n + 1) >> 1) - g0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < np;
         i += (int64_t)gridDim.x * 256) {
        double a, b;
        uniforms2(g0 + i, seed, offset, a, b);
        store_pair(out, n, e0, g0 + i, a, b);
    }
}

__global__ void __launch_bounds__(256)
rng_normal_kernel(double *out, int64_t n, uint64_t seed, uint64_t offset, int64_t e0)
{
    const int64_t g0 = e0 >> 1;
    const int64_t np = ((e0 + n + 1) >> 1) - g0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < np;
         i += (int64_t)gridDim.x * 256) {
        double a, b;
        normals2(g0 + i, seed, offset, a, b);
        store_pair(out, n, e0, g0 + i, a, b);
    }
}

// The uniform draws of an HMC transition ride along (rng_normal_zig_uniform_kernel): [C]
// acceptance draws next to [C x D] momenta are not worth a launch of their own.
struct UniformTail {
    double *out;
    int64_t n;
    uint64_t offset;
    int64_t e0;
};

template <bool TAIL>
__device__ inline void normal_zig_body(double *out, int64_t n, uint64_t seed, uint64_t offset, int64_t e0,
                                       const UniformTail u)
{
    if (TAIL) {
        const int64_t g0 = u.e0 >> 1;
        const int64_t np = ((u.e0 + u.n + 1) >> 1) - g0;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < np; i += (int64_t)gridDim.x * 256) {
            double a, b;
            uniforms2(g0 + i, seed, u.offset, a, b);
            store_pair(u.out, u.n, u.e0, g0 + i, a, b);
        }
    }
    __shared__ double zx[ZIG_C + 1];
    __shared__ double zr[ZIG_C];
    for (int k = threadIdx.x; k <= ZIG_C; k += 256) zx[k] = ZIG_X[k];
    for (int k = threadIdx.x; k < ZIG_C; k += 256) zr[k] = ZIG_RATIO[k];
    __syncthreads();
    const int64_t g0 = e0 >> 1;                     // first Philox block = global output pair
    const int64_t np = ((e0 + n + 1) >> 1) - g0;    // blocks the window [e0, e0 + n) touches
    // 16-byte stores need the pairs aligned with the buffer: even window start
    const bool vec = !(e0 & 1) && !((uintptr_t)out & 15);
    // a workgroup covers 1024 consecutive pairs per round: lane t owns pairs
    // base + t, base + 256 + t, ... so that stores are 16 B per lane, coalesced
    for (int64_t base = g0 + (int64_t)blockIdx.x * 1024; base < g0 + np;
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
            const int64_t l = 2 * i - e0;
            if (vec && l + 1 < n) {
                typedef double v2d __attribute__((ext_vector_type(2)));
                v2d w;
                w.x = val[2 * b];
                w.y = val[2 * b + 1];
                *reinterpret_cast<v2d *>(out + l) = w;
            } else {
                store_pair(out, n, e0, i, val[2 * b], val[2 * b + 1]);
            }
        }
    }
}

__global__ void __launch_bounds__(256)
rng_normal_zig_kernel(double *out, int64_t n, uint64_t seed, uint64_t offset, int64_t e0)
{
    normal_zig_body<false>(out, n, seed, offset, e0, UniformTail{nullptr, 0, 0, 0});
}

__global__ void __launch_bounds__(256)
rng_normal_zig_uniform_kernel(double *out, int64_t n, uint64_t seed, uint64_t offset, int64_t e0,
                              const UniformTail u)
{
    normal_zig_body<true>(out, n, seed, offset, e0, u);
}

__global__ void __launch_bounds__(256)
rng_gamma_kernel(double *out, int64_t n, double shape, uint64_t seed, uint64_t offset, int64_t e0)
{
    for (int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x; l < n;
         l += (int64_t)gridDim.x * 256)
        out[l] = gamma_elem(e0 + l, shape, seed, offset);       // global element e0 + l
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
                                        uint64_t offset, int64_t elem_offset, void *stream)
{
    if (n < 0 || elem_offset < 0) return fail(BINF_E_ARG, "rng_uniform: negative size / offset");
    if (n == 0) return 0;
    if (!out) return fail(BINF_E_ARG, "rng_uniform: null buffer");
    rng_uniform_kernel<<<dim3(rng_grid((n + 2) / 2)), 256, 0, (hipStream_t)stream>>>(out, n, seed, offset,
                                                                                      elem_offset);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rng_uniform launch");
    return 0;
}

extern "C" int32_t binf_rng_normal_f64(double *out, int64_t n, uint64_t seed,
                                       uint64_t offset, int64_t elem_offset, void *stream)
{
    if (n < 0 || elem_offset < 0) return fail(BINF_E_ARG, "rng_normal: negative size / offset");
    if (n == 0) return 0;
    if (!out) return fail(BINF_E_ARG, "rng_normal: null buffer");
    rng_normal_kernel<<<dim3(rng_grid((n + 2) / 2)), 256, 0, (hipStream_t)stream>>>(out, n, seed, offset,
                                                                                     elem_offset);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rng_normal launch");
    return 0;
}

extern "C" int32_t binf_rng_normal_zig_uniform_f64(double *normals, int64_t n_normals,
                                                   double *uniforms, int64_t n_uniforms,
                                                   uint64_t seed, uint64_t offset_normals,
                                                   uint64_t offset_uniforms,
                                                   int64_t elem_offset_normals,
                                                   int64_t elem_offset_uniforms, void *stream)
{
    if (n_normals < 0 || n_uniforms < 0 || elem_offset_normals < 0 || elem_offset_uniforms < 0)
        return fail(BINF_E_ARG, "rng_normal_zig_uniform: negative size / offset");
    if (offset_normals >> 48) return fail(BINF_E_ARG, "rng_normal_zig_uniform: offset must be < 2^48");
    if (offset_normals == offset_uniforms)
        return fail(BINF_E_ARG, "rng_normal_zig_uniform: the two streams need different offsets");
    if (n_normals == 0 && n_uniforms == 0) return 0;
    if ((n_normals > 0 && !normals) || (n_uniforms > 0 && !uniforms))
        return fail(BINF_E_ARG, "rng_normal_zig_uniform: null buffer");
    if (n_normals == 0)
        return binf_rng_uniform_f64(uniforms, n_uniforms, seed, offset_uniforms, elem_offset_uniforms, stream);
    UniformTail u;
    u.out = uniforms; u.n = n_uniforms; u.offset = offset_uniforms; u.e0 = elem_offset_uniforms;
    rng_normal_zig_uniform_kernel<<<dim3(rng_grid((n_normals + 9) / 8)), 256, 0, (hipStream_t)stream>>>(
        normals, n_normals, seed, offset_normals, elem_offset_normals, u);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rng_normal_zig_uniform launch");
    return 0;
}

extern "C" int32_t binf_rng_gamma_f64(double *out, int64_t n, double shape, uint64_t seed,
                                      uint64_t offset, int64_t elem_offset, void *stream)
{
    if (n < 0 || elem_offset < 0 || !(shape > 0.0))
        return fail(BINF_E_ARG, "rng_gamma: need n >= 0, elem_offset >= 0 and shape > 0");
    if (n == 0) return 0;
    if (!out) return fail(BINF_E_ARG, "rng_gamma: null buffer");
    rng_gamma_kernel<<<dim3(rng_grid(n)), 256, 0, (hipStream_t)stream>>>(out, n, shape, seed, offset,
                                                                         elem_offset);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rng_gamma launch");
    return 0;
}

extern "C" int32_t binf_rng_normal_zig_f64(double *out, int64_t n, uint64_t seed,
                                           uint64_t offset, int64_t elem_offset, void *stream)
{
    if (n < 0 || elem_offset < 0) return fail(BINF_E_ARG, "rng_normal_zig: negative size / offset");
    if (offset >> 48) return fail(BINF_E_ARG, "rng_normal_zig: offset must be < 2^48");
    if (n == 0) return 0;
    if (!out) return fail(BINF_E_ARG, "rng_normal_zig: null buffer");
    rng_normal_zig_kernel<<<dim3(rng_grid((n + 9) / 8)), 256, 0, (hipStream_t)stream>>>(out, n, seed, offset,
                                                                                         elem_offset);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rng_normal_zig launch");
    return 0;
}

// LDS pipe: conflict-free ds_add_f64 against ds_read_b64, 16 waves per CU (would accumulating the
// partner sums of the pair-distance force with LDS atomics beat rotating them through DPP moves?
// DESIGN.md section 4.4).  hipcc --offload-arch=gfx950 -O3 -o /tmp/ldsatom scripts/lds_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
// throughput of conflict-free LDS f64 atomic adds vs plain LDS reads, 16 waves per workgroup, 1 per CU
__global__ void __launch_bounds__(1024) k_atomic(double *out, int iters)
{
    __shared__ double buf[16][3][128];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (int a = 0; a < 3; ++a) { buf[w][a][l] = 0.0; buf[w][a][l + 64] = 0.0; }
    __syncthreads();
    double v = 1.0 + l;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            __hip_atomic_fetch_add(&buf[w][0][l + k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&buf[w][1][l + k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&buf[w][2][l + k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    out[blockIdx.x * 1024 + threadIdx.x] = buf[w][0][l];
}
__global__ void __launch_bounds__(1024) k_read(double *out, int iters)
{
    __shared__ double buf[16][3][128];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (int a = 0; a < 3; ++a) { buf[w][a][l] = 1.0; buf[w][a][l + 64] = 2.0; }
    __syncthreads();
    double s = 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            s += buf[w][0][l + k] + buf[w][1][l + k] + buf[w][2][l + k];
        }
        asm volatile("" : "+v"(s));
    }
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}
int main()
{
    double *out; hipMalloc(&out, 256 * 1024 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (which == 0) k_atomic<<<256, 1024>>>(out, 200); else k_read<<<256, 1024>>>(out, 200);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per CU: 16 waves x 200 x 96 LDS instructions
            printf("%s: %.3f ms -> %.1f ns per wave-instruction per CU (16 waves share the pipe)\n",
                   which == 0 ? "ds_add_f64" : "ds_read_b64", ms, ms * 1e6 / (16.0 * 200 * 96));
        }
    }
    return 0;
}

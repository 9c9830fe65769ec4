// FP64 MFMA ceiling on gfx950: v_mfma_f64_16x16x4_f64 back to back, no memory.
// NCH independent accumulator chains per wave, 1..4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma64bench.hip -o scripts/mfma64bench
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));

template <int NCH>
__global__ void __launch_bounds__(256) mm(double *out, double seed, int iters)
{
    v4d acc[NCH];
    double a = seed * (threadIdx.x + 1) * 1e-3, b = seed * (threadIdx.x + 3) * 1e-3;
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int l = 0; l < iters; ++l) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NCH>
static void run(int wps, int iters)
{
    const int blocks = 256 * wps;
    double *out;
    (void)hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) mm<NCH><<<blocks, 256>>>(out, 1.0, iters);
    (void)hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) mm<NCH><<<blocks, 256>>>(out, 1.0, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double t = ms * 1e-3 / reps;
    const double n = (double)iters * NCH * wps;             // MFMAs per SIMD
    printf("chains/wave=%d waves/SIMD=%d: %.1f us, %.1f TFLOP/s, %.1f cycles per MFMA at 2.4 GHz\n",
           NCH, wps, t * 1e6, n * 1024 * 2048 / t * 1e-12, t * 2.4e9 / n);
    (void)hipFree(out);
}

int main()
{
    const int iters = 20000;
    run<1>(1, iters); run<2>(1, iters); run<4>(1, iters);
    run<1>(2, iters); run<2>(2, iters); run<4>(2, iters);
    run<2>(3, iters); run<4>(3, iters); run<4>(4, iters);
    return 0;
}

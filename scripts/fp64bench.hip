// FP64 VALU ceiling for the EXACT-mode leapfrog arithmetic on gfx950: the same
// mul/add sequence as hmc_gauss_persist_kernel's step loop, no memory traffic.
// Reports wave-instruction issue rate as an effective clock (instructions x 4
// cycles / SIMD / time) for 1..4 waves per SIMD and for random vs zero data
// (the latter shows how much of the limit is power, not issue slots).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/fp64bench.hip -o scripts/fp64bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int E, bool FMA>
__global__ void __launch_bounds__(256) leap(double *out, double seed, double dt, int iters)
{
    double q[E], p[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        q[i] = seed * (double)(threadIdx.x * 31 + i * 7 + blockIdx.x + 1) * 1.2345678901234e-3;
        p[i] = seed * (double)(threadIdx.x * 17 + i * 3 + blockIdx.x + 2) * 0.9876543210987e-3;
    }
    for (int l = 0; l < iters; ++l) {
#pragma unroll
        for (int i = 0; i < E; ++i) {
            if (FMA) {
                q[i] = __builtin_fma(p[i], dt, q[i]);
                p[i] = __builtin_fma(-dt, q[i], p[i]);
            } else {
                q[i] = q[i] + p[i] * dt;
                p[i] = p[i] - dt * q[i];
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < E; ++i) s += q[i] + p[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int E, bool FMA>
static void run(int waves_per_simd, double seed, int iters)
{
    const int blocks = 256 * waves_per_simd;             // 256 CUs x 4 SIMDs, 4 waves per block
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) leap<E, FMA><<<blocks, 256>>>(out, seed, 0.05, iters);
    hipEventRecord(a);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) leap<E, FMA><<<blocks, 256>>>(out, seed, 0.05, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double t = ms * 1e-3 / reps;
    const double winst = (double)iters * E * (FMA ? 2 : 4) * waves_per_simd;   // per SIMD
    printf("E=%d %s waves/SIMD=%d data=%s: %.1f us, %.2f T lane-ops/s, effective issue clock %.2f GHz\n",
           E, FMA ? "fma" : "exact", waves_per_simd, seed == 0.0 ? "zero" : "random", t * 1e6,
           winst * 64 * 1024 / t * 1e-12, winst * 4 / t * 1e-9);
    hipFree(out);
}

int main()
{
    const int iters = 20000;
    for (double seed : {1.0, 0.0})
        for (int w : {1, 2, 3, 4, 8}) run<8, false>(w, seed, iters);
    run<8, true>(4, 1.0, iters);
    run<8, true>(4, 0.0, iters);
    run<16, false>(4, 1.0, iters);
    return 0;
}

// Cost of the FP64 and FP32 reciprocal-square-root seeds on gfx950 (no memory
// traffic): E independent chains of  x = op(x) * a + b  per lane, one wave per
// SIMD up to four.  Reports SIMD cycles per wave-instruction of `op` after
// subtracting nothing (the FMA beside it is 4 cycles): the question behind
// pair_weight() in binf_amd/csrc/pairdist.hip.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/rsqbench.hip -o scripts/rsqbench
#include <hip/hip_runtime.h>
#include <cstdio>

// KIND 0: v_rsq_f64; 1: v_rsq_f32 on the converted value, converted back;
// 2: FMA only (baseline); 3: v_rcp_f64; 4: v_sqrt_f64
template <int KIND, int E>
__global__ void __launch_bounds__(256) chain(double *out, double seed, int iters)
{
    double x[E];
#pragma unroll
    for (int i = 0; i < E; ++i) x[i] = 1.0 + seed * (double)(threadIdx.x * 31 + i * 7 + blockIdx.x + 1) * 1e-3;
    for (int l = 0; l < iters; ++l) {
#pragma unroll
        for (int i = 0; i < E; ++i) {
            double r;
            if (KIND == 0) r = __builtin_amdgcn_rsq(x[i]);
            else if (KIND == 1) r = (double)__builtin_amdgcn_rsqf((float)x[i]);
            else if (KIND == 3) r = __builtin_amdgcn_rcp(x[i]);
            else if (KIND == 4) r = __builtin_amdgcn_sqrt(x[i]);
            else r = x[i];
            x[i] = __builtin_fma(r, 0.75, 1.25);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < E; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int E>
static void run(const char *name, int waves_per_simd, int iters)
{
    const int blocks = 256 * waves_per_simd;
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) chain<KIND, E><<<blocks, 256>>>(out, 1.0, iters);
    hipEventRecord(a);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) chain<KIND, E><<<blocks, 256>>>(out, 1.0, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double t = ms * 1e-3 / reps;
    const double ops = (double)iters * E * waves_per_simd;      // wave-level op+fma pairs per SIMD
    printf("%-22s E=%d waves/SIMD=%d: %8.1f us  %.1f ns per (op + fma) per SIMD = %.0f cycles at 2.4 GHz\n",
           name, E, waves_per_simd, t * 1e6, t / ops * 1e9, t / ops * 2.4e9);
    hipFree(out);
}

int main()
{
    const int it = 2000;
    for (int w = 1; w <= 4; w *= 2) {
        run<2, 4>("fma only", w, it);
        run<0, 4>("v_rsq_f64", w, it);
        run<1, 4>("cvt + v_rsq_f32 + cvt", w, it);
        run<3, 4>("v_rcp_f64", w, it);
        run<4, 4>("v_sqrt_f64", w, it);
    }
    return 0;
}

// What FP64 MFMA rate does an MI355X SUSTAIN?  (evidence for the C3 / C4 roofline)
//
// v_mfma_f64_16x16x4_f64 occupies a SIMD's matrix pipe for 64 cycles (2048 flop),
// so 1024 SIMDs x 2048 / 64 x 2.4 GHz = 78.6 TFLOP/s on the datasheet.  Under a
// dense FP64-MFMA load the chip does not hold 2.4 GHz (DVFS give-back,
// MI355X_MICROARCH.md): what a kernel can reach is duty x clock(duty, data), and
// the clock falls as the duty rises.  This sweeps that curve with no memory traffic:
//
//   * operands in registers, 8 rotating (a, b) pairs of RANDOM doubles (zero / constant
//     operands clock higher and miss the point -- also measured, as a contrast);
//   * NCH independent accumulator chains per wave, WPS waves per SIMD (LDS request pins
//     the workgroups per CU);
//   * after every MFMA a pad of NOP x `s_nop 7` (8 idle cycles each) and VF independent
//     v_fma_f64: the duty cycle of the matrix pipe falls as the pad grows;
//   * every configuration runs back to back for SETTLE seconds first, then REPS launches
//     are timed with HIP events; s_memtime / s_memrealtime stamps around the loop give the
//     in-kernel shader clock (delta memtime / delta realtime x 100 MHz, median over
//     workgroups) and the cycles per MFMA the wave really saw.
//
// Prints one JSON object per configuration.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma64_duty.hip -o scripts/mfma64_duty
//   scripts/mfma64_duty [settle_seconds=1.5] > gpurun_out/mfma64_duty.jsonl
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

template <int NCH, int NOP, int VF>
__global__ void __launch_bounds__(256)
duty_kernel(double *out, unsigned long long *stamps, const double *ops, int iters)
{
    extern __shared__ double pin[];            // only to bound the workgroups per CU
    v4d acc[NCH];
    double a[8], b[8];
    double f[VF > 0 ? VF : 1];
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = ops[(j * 2 + 0) * 256 + tid];
        b[j] = ops[(j * 2 + 1) * 256 + tid];
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int v = 0; v < (VF > 0 ? VF : 1); ++v) f[v] = a[v & 7];
    if (tid == 1 << 20) pin[0] = 0.0;          // keeps the LDS request alive

    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int l = 0; l < iters; ++l) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], b[j], acc[c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < NOP; ++n) asm volatile("s_nop 7");
#pragma unroll
                for (int v = 0; v < VF; ++v)
                    asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f[v]) : "v"(a[j]), "v"(b[(j + v) & 7]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);

    double s = 0.0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
#pragma unroll
    for (int v = 0; v < VF; ++v) s += f[v];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (tid == 0) {                            // stamps go to a buffer nothing else reads
        stamps[(size_t)blockIdx.x * 2 + 0] = t1 - t0;
        stamps[(size_t)blockIdx.x * 2 + 1] = r1 - r0;
    }
}

static double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.0 : v[v.size() / 2];
}

template <int NCH, int NOP, int VF>
static void run(int wps, const char *data, const double *ops_dev, double settle_s)
{
    const int blocks = 256 * wps;              // one workgroup (4 waves) per CU and wave slot
    // LDS request: 160 KiB per CU shared by wps workgroups (64 KiB is the per-workgroup limit)
    size_t lds = wps == 1 ? 64 * 1024 : (wps == 2 ? 64 * 1024 : (wps == 3 ? 48 * 1024 : 36 * 1024));
    double *out;
    unsigned long long *stamps;
    (void)hipMalloc(&out, sizeof(double) * blocks * 256);
    (void)hipMalloc(&stamps, sizeof(unsigned long long) * blocks * 2);
    auto kern = duty_kernel<NCH, NOP, VF>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // ~4 ms per launch at full duty: iters x 8 x NCH MFMAs x 64 cycles x wps at ~2 GHz
    const int iters = std::max(64, (int)(4e-3 * 2.0e9 / (8.0 * NCH * 64.0 * wps)));
    auto t_s = std::chrono::steady_clock::now();
    int settled = 0;
    do {
        for (int r = 0; r < 8; ++r) kern<<<blocks, 256, lds>>>(out, stamps, ops_dev, iters);
        (void)hipDeviceSynchronize();
        settled += 8;
    } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_s).count() < settle_s);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int reps = 16;
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) kern<<<blocks, 256, lds>>>(out, stamps, ops_dev, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    std::vector<unsigned long long> h(blocks * 2);
    (void)hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * blocks * 2, hipMemcpyDeviceToHost);
    std::vector<double> clk, cyc;
    for (int i = 0; i < blocks; ++i)
        if (h[2 * i + 1] > 0) {
            clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);      // GHz (100 MHz ticks)
            cyc.push_back((double)h[2 * i]);
        }
    const double t = ms * 1e-3 / reps;
    const double mfma_wave = (double)iters * 8 * NCH;
    const double tf = mfma_wave * 2048.0 * 4.0 * blocks / t * 1e-12;
    const double cyc_per_mfma_simd = median(cyc) / (mfma_wave * wps);            // matrix-pipe period
    printf("{\"data\": \"%s\", \"chains_per_wave\": %d, \"waves_per_simd\": %d, \"nop8_per_mfma\": %d, "
           "\"vfma_per_mfma\": %d, \"launch_us\": %.1f, \"TFLOPs\": %.2f, \"frac_of_78.6\": %.3f, "
           "\"clock_GHz_in_kernel\": %.3f, \"cycles_per_mfma_per_simd\": %.1f, \"matrix_pipe_duty\": %.3f, "
           "\"TFLOPs_at_2.4GHz_same_cycles\": %.2f, \"settle_launches\": %d, \"iters\": %d, \"hip_error\": %d}\n",
           data, NCH, wps, NOP, VF, t * 1e6, tf, tf / 78.6, median(clk), cyc_per_mfma_simd,
           64.0 / cyc_per_mfma_simd, 78.6 * 64.0 / cyc_per_mfma_simd, settled, iters, (int)err);
    fflush(stdout);
    (void)hipFree(out);
    (void)hipFree(stamps);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

int main(int argc, char **argv)
{
    const double settle = argc > 1 ? atof(argv[1]) : 1.5;
    std::vector<double> rnd(16 * 256), zero(16 * 256, 0.0);
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    for (auto &x : rnd) {                       // xorshift: uniform(-1, 1) / 16, full mantissas
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        x = ((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0) * 0.0625;
    }
    double *d_rnd, *d_zero;
    (void)hipMalloc(&d_rnd, sizeof(double) * rnd.size());
    (void)hipMalloc(&d_zero, sizeof(double) * zero.size());
    (void)hipMemcpy(d_rnd, rnd.data(), sizeof(double) * rnd.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_zero, zero.data(), sizeof(double) * zero.size(), hipMemcpyHostToDevice);

    // bare loops: chains per wave and waves per SIMD, random data
    run<1, 0, 0>(1, "random", d_rnd, settle);
    run<2, 0, 0>(1, "random", d_rnd, settle);
    run<4, 0, 0>(1, "random", d_rnd, settle);
    run<4, 0, 0>(2, "random", d_rnd, settle);
    run<4, 0, 0>(3, "random", d_rnd, settle);
    // the same on zeros: the clock the chip holds when the data cost nothing
    run<4, 0, 0>(1, "zero", d_zero, settle);
    run<4, 0, 0>(3, "zero", d_zero, settle);
    // duty sweep, one wave per SIMD (the pad is idle time of the matrix pipe once it
    // exceeds the 64 cycles the MFMA holds it)
    run<4, 7, 0>(1, "random", d_rnd, settle);
    run<4, 8, 0>(1, "random", d_rnd, settle);
    run<4, 9, 0>(1, "random", d_rnd, settle);
    run<4, 10, 0>(1, "random", d_rnd, settle);
    run<4, 11, 0>(1, "random", d_rnd, settle);
    run<4, 12, 0>(1, "random", d_rnd, settle);
    run<4, 14, 0>(1, "random", d_rnd, settle);
    run<4, 16, 0>(1, "random", d_rnd, settle);
    // pads of FP64 VALU work instead of idle cycles (what a real kernel puts beside its MFMAs;
    // the gradient kernel issues ~4.2 VALU instructions per MFMA)
    run<4, 0, 4>(1, "random", d_rnd, settle);
    run<4, 0, 8>(1, "random", d_rnd, settle);
    run<4, 0, 4>(3, "random", d_rnd, settle);
    run<4, 0, 12>(1, "random", d_rnd, settle);
    run<4, 0, 16>(1, "random", d_rnd, settle);
    run<4, 4, 8>(1, "random", d_rnd, settle);
    return 0;
}

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

// KIND of the VF filler instructions after every MFMA: 0 v_fma_f64, 1 v_fma_f32, 2 v_mov_b32,
// 3 v_add_u32, 4 v_cndmask_b32, 5 ds_read_b64 (LDS), 6 v_add_f64, 7 v_mul_f64
template <int NCH, int NOP, int VF, int KIND = 0>
__global__ void __launch_bounds__(256)
duty_kernel(double *out, unsigned long long *stamps, const double *ops, int iters)
{
    extern __shared__ double pin[];            // only to bound the workgroups per CU
    v4d acc[NCH];
    double a[8], b[8];
    constexpr int VFN = VF > 0 ? VF : 1;
    double f[VFN];
    float g32[VFN];
    unsigned i32[VFN];
    const int tid = threadIdx.x;
    const unsigned ldsaddr = (unsigned)(tid * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = ops[(j * 2 + 0) * 256 + tid];
        b[j] = ops[(j * 2 + 1) * 256 + tid];
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int v = 0; v < VFN; ++v) {
        f[v] = a[v & 7];
        g32[v] = (float)b[v & 7];
        i32[v] = (unsigned)tid + v;
    }
    pin[tid] = a[0];                           // keeps the LDS request alive (and readable)
    __syncthreads();

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
                for (int n = 0; n < NOP; ++n) asm volatile("s_nop 1");      // 8 cycles each
#pragma unroll
                for (int v = 0; v < VF; ++v) {
                    if (KIND == 0)
                        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f[v]) : "v"(a[j]), "v"(b[(j + v) & 7]));
                    else if (KIND == 6)
                        asm volatile("v_add_f64 %0, %1, %0" : "+v"(f[v]) : "v"(a[j]));
                    else if (KIND == 7)
                        asm volatile("v_mul_f64 %0, %1, %0" : "+v"(f[v]) : "v"(b[(j + v) & 7]));
                    else if (KIND == 1)
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(g32[v]) : "v"(g32[(v + 1) % VFN]), "v"(g32[(v + 2) % VFN]));
                    else if (KIND == 2)
                        asm volatile("v_mov_b32 %0, %1" : "=v"(g32[v]) : "v"(g32[(v + 1) % VFN]));
                    else if (KIND == 3)
                        asm volatile("v_add_u32 %0, %1, %0" : "+v"(i32[v]) : "v"(i32[(v + 1) % VFN]));
                    else if (KIND == 4)
                        asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i32[v]) : "v"(i32[(v + 1) % VFN]), "v"(i32[(v + 2) % VFN]) : );
                    else if (KIND == 5)
                        asm volatile("ds_read_b64 %0, %1" : "=v"(f[v]) : "v"(ldsaddr));
                }
                if (KIND == 5 && VF > 0) asm volatile("s_waitcnt lgkmcnt(0)");
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
    for (int v = 0; v < VF; ++v) s += f[v] + (double)g32[v] + (double)i32[v];
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

static const char *KIND_NAME[] = {"v_fma_f64", "v_fma_f32", "v_mov_b32", "v_add_u32", "v_cndmask_b32",
                                  "ds_read_b64", "v_add_f64", "v_mul_f64"};

template <int NCH, int NOP, int VF, int KIND = 0>
static void run(int wps, const char *data, const double *ops_dev, double settle_s)
{
    const int blocks = 256 * wps;              // one workgroup (4 waves) per CU and wave slot
    // LDS request: 160 KiB per CU shared by wps workgroups (64 KiB is the per-workgroup limit)
    size_t lds = wps == 1 ? 64 * 1024 : (wps == 2 ? 64 * 1024 : (wps == 3 ? 48 * 1024 : 36 * 1024));
    double *out;
    unsigned long long *stamps;
    (void)hipMalloc(&out, sizeof(double) * blocks * 256);
    (void)hipMalloc(&stamps, sizeof(unsigned long long) * blocks * 2);
    auto kern = duty_kernel<NCH, NOP, VF, KIND>;
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
    // matrix-pipe period of a SIMD: wall time x in-kernel clock / MFMAs per SIMD
    const double cyc_per_mfma_simd = t * median(clk) * 1e9 / (mfma_wave * blocks * 4.0 / 1024.0);
    const double wave_cyc_per_mfma = median(cyc) / mfma_wave;   // what one wave saw (its own stamps)
    printf("{\"data\": \"%s\", \"chains_per_wave\": %d, \"waves_per_simd\": %d, \"nop8_per_mfma\": %d, "
           "\"filler\": \"%s\", \"wave_cycles_per_mfma\": %.1f, \"fillers_per_mfma\": %d, \"launch_us\": %.1f, \"TFLOPs\": %.2f, \"frac_of_78.6\": %.3f, "
           "\"clock_GHz_in_kernel\": %.3f, \"cycles_per_mfma_per_simd\": %.1f, \"matrix_pipe_duty\": %.3f, "
           "\"TFLOPs_at_2.4GHz_same_cycles\": %.2f, \"settle_launches\": %d, \"iters\": %d, \"hip_error\": %d}\n",
           data, NCH, wps, NOP, KIND_NAME[KIND], wave_cyc_per_mfma, VF, t * 1e6, tf, tf / 78.6, median(clk), cyc_per_mfma_simd,
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
    run<4, 0, 0>(1, "random", d_rnd, settle);
    run<4, 0, 0>(2, "random", d_rnd, settle);
    run<4, 0, 0>(3, "random", d_rnd, settle);
    run<4, 0, 0>(4, "random", d_rnd, settle);
    run<4, 0, 0>(2, "zero", d_zero, settle);
    // idle pads (8 cycles each): where does the matrix pipe start to starve?
    run<4, 4, 0>(1, "random", d_rnd, settle);
    run<4, 6, 0>(1, "random", d_rnd, settle);
    run<4, 7, 0>(1, "random", d_rnd, settle);
    run<4, 8, 0>(1, "random", d_rnd, settle);
    run<4, 10, 0>(1, "random", d_rnd, settle);
    // what does ONE filler instruction beside an FP64 MFMA cost, by kind?  4 and 8 per MFMA,
    // one and three waves per SIMD
#define KINDS(W)                                                                         \
    run<4, 0, 4, 0>(W, "random", d_rnd, settle); run<4, 0, 8, 0>(W, "random", d_rnd, settle); \
    run<4, 0, 4, 6>(W, "random", d_rnd, settle); run<4, 0, 8, 6>(W, "random", d_rnd, settle); \
    run<4, 0, 4, 7>(W, "random", d_rnd, settle); run<4, 0, 8, 7>(W, "random", d_rnd, settle); \
    run<4, 0, 4, 1>(W, "random", d_rnd, settle); run<4, 0, 8, 1>(W, "random", d_rnd, settle); \
    run<4, 0, 4, 2>(W, "random", d_rnd, settle); run<4, 0, 8, 2>(W, "random", d_rnd, settle); \
    run<4, 0, 4, 3>(W, "random", d_rnd, settle); run<4, 0, 8, 3>(W, "random", d_rnd, settle); \
    run<4, 0, 4, 4>(W, "random", d_rnd, settle); run<4, 0, 8, 4>(W, "random", d_rnd, settle); \
    run<4, 0, 2, 5>(W, "random", d_rnd, settle); run<4, 0, 4, 5>(W, "random", d_rnd, settle);
    KINDS(1)
    KINDS(3)
    return 0;
}

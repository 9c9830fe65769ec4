// Micro-benchmark (development aid, not product): memory floor of the C2
// traffic shape -- per chain read 2 x 8 KiB (q0, p0), write 8 KiB -- for
// different access patterns.  hipcc --offload-arch=gfx950 -O3 membench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../binf_amd/csrc/api.hip"
#include "../binf_amd/csrc/hmc_gauss.hip"
typedef double d2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int D = 1024;

// A: ownership layout, 8-byte accesses, 64-byte segments (what the kernel does)
__global__ void __launch_bounds__(256) k_strided8(const double *q0, const double *p0, double *out, int C)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= C) return;
    const long base = wave * D + (lane >> 3) * 128 + (lane & 7);
    double q[16], p[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) q[t] = q0[base + 8 * t];
#pragma unroll
    for (int t = 0; t < 16; ++t) p[t] = p0[base + 8 * t];
#pragma unroll
    for (int t = 0; t < 16; ++t) out[base + 8 * t] = q[t] + p[t];
}

// B: natural layout, 16-byte accesses, fully coalesced
template <bool NT>
__global__ void __launch_bounds__(256) k_coal16(const d2v *q0, const d2v *p0, d2v *out, int C)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= C) return;
    const long base = wave * (D / 2) + lane;
    d2v q[8], p[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) q[t] = q0[base + 64 * t];
#pragma unroll
    for (int t = 0; t < 8; ++t)
        p[t] = NT ? __builtin_nontemporal_load(&p0[base + 64 * t]) : p0[base + 64 * t];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        d2v r = q[t] + p[t];
        out[base + 64 * t] = r;
    }
}

// C: 16-byte accesses where each lane owns 2 accumulators of a leaf: 4 lanes
// cover a 64-byte segment (ownership-compatible, 32 lanes per chain)
__global__ void __launch_bounds__(256) k_strided16(const d2v *q0, const d2v *p0, d2v *out, int C)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long chain = wave * 2 + (lane >> 5);
    if (chain >= C) return;
    const int sl = lane & 31;
    const long base = chain * (D / 2) + (sl >> 2) * 64 + (sl & 3);
    d2v q[16], p[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) q[t] = q0[base + 4 * t];
#pragma unroll
    for (int t = 0; t < 16; ++t) p[t] = p0[base + 4 * t];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        d2v r = q[t] + p[t];
        out[base + 4 * t] = r;
    }
}

// D: as A plus W leapfrog steps, an optional wave reduction + exp-dependent
// store (RED), optional 1-byte flag store / counter update per chain (FLAGS)
template <int W, bool RED, bool FLAGS>
__global__ void __launch_bounds__(256) k_work(const double *q0, const double *p0, double *out,
                                              unsigned char *flag, long *cnt, const double *u, int C)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= C) return;
    const long base = wave * D + (lane >> 3) * 128 + (lane & 7);
    const double uu = u[wave];
    double q[16], p[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) { q[t] = q0[base + 8 * t]; p[t] = p0[base + 8 * t]; }
    const double dt = 0.05;
    for (int s = 0; s < W; ++s) {
#pragma unroll
        for (int t = 0; t < 16; ++t) { q[t] = q[t] + p[t] * dt; p[t] = p[t] - dt * q[t]; }
    }
    bool acc = true;
    if (RED) {
        double r = p[0] * p[0];
#pragma unroll
        for (int t = 1; t < 16; ++t) r = r + p[t] * p[t];
        for (int m = 1; m < 64; m <<= 1) r = r + __shfl_xor(r, m, 64);
        acc = uu < exp(-r * 1e-9);
    }
    if (FLAGS && lane == 0) { flag[wave] = acc; if (acc) cnt[wave] += 1; }
    if (acc) {
#pragma unroll
        for (int t = 0; t < 16; ++t) out[base + 8 * t] = q[t];
    }
}

// E: closer to the product kernel.  REDB: numpy-order reductions of q**2 and
// p**2 BEFORE the trajectory (E_before) and again after it; GM: element-group-
// major order (4 elements at a time run the whole trajectory).
__device__ inline double wave_tree(double r)
{
    for (int m = 1; m < 64; m <<= 1) r = r + __shfl_xor(r, m, 64);
    return r;
}
template <bool REDB, bool GM>
__global__ void __launch_bounds__(256) k_real(const double *q0, const double *p0, double *out,
                                              unsigned char *flag, long *cnt, const double *u, int C, int W)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= C) return;
    const long base = wave * D + (lane >> 3) * 128 + (lane & 7);
    const double uu = u[wave];
    double q[16], p[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) { q[t] = q0[base + 8 * t]; p[t] = p0[base + 8 * t]; }
    const double dt = 0.05, hdt = 0.5 * dt;
    double sqb = 0, spb = 0, sqa = 0, spa = 0;
    if (GM) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { int t = 4 * g + i; sqb = t ? sqb + q[t] * q[t] : q[t] * q[t]; spb = t ? spb + p[t] * p[t] : p[t] * p[t]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) { int t = 4 * g + i; p[t] = p[t] - hdt * q[t]; }
            for (int s = 0; s < W - 1; ++s) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { int t = 4 * g + i; q[t] = q[t] + p[t] * dt; p[t] = p[t] - dt * q[t]; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { int t = 4 * g + i; q[t] = q[t] + p[t] * dt; p[t] = p[t] - hdt * q[t]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) { int t = 4 * g + i; sqa = t ? sqa + q[t] * q[t] : q[t] * q[t]; spa = t ? spa + p[t] * p[t] : p[t] * p[t]; }
        }
        sqb = wave_tree(sqb); spb = wave_tree(spb);
    } else {
        if (REDB) {
#pragma unroll
            for (int t = 0; t < 16; ++t) { sqb = t ? sqb + q[t] * q[t] : q[t] * q[t]; spb = t ? spb + p[t] * p[t] : p[t] * p[t]; }
            sqb = wave_tree(sqb); spb = wave_tree(spb);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) p[t] = p[t] - hdt * q[t];
        for (int s = 0; s < W - 1; ++s) {
#pragma unroll
            for (int t = 0; t < 16; ++t) { q[t] = q[t] + p[t] * dt; p[t] = p[t] - dt * q[t]; }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) { q[t] = q[t] + p[t] * dt; p[t] = p[t] - hdt * q[t]; }
#pragma unroll
        for (int t = 0; t < 16; ++t) { sqa = t ? sqa + q[t] * q[t] : q[t] * q[t]; spa = t ? spa + p[t] * p[t] : p[t] * p[t]; }
    }
    sqa = wave_tree(sqa); spa = wave_tree(spa);
    const double Eb = 0.5 * sqb + 0.5 * spb, Ea = 0.5 * sqa + 0.5 * spa;
    double x = -(Ea - Eb); x = x < -308.0 ? -308.0 : x; x = x > 709.0 ? 709.0 : x;
    const bool acc = uu < exp(x);
    if (lane == 0) { flag[wave] = acc; if (acc) cnt[wave] += 1; }
    if (acc) {
#pragma unroll
        for (int t = 0; t < 16; ++t) out[base + 8 * t] = q[t];
    }
}
template <bool REDB, bool GM>
static void run_real(const char *name, int W, int C, int K, int P, double *qa, double *qb, std::vector<double *> &pool,
                     unsigned char *flag, long *cnt, double *u, hipEvent_t e0, hipEvent_t e1, double bytes)
{
    for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1) CK(hipEventRecord(e0));
        for (int i = 0; i < K; ++i) {
            double *src = (i & 1) ? qb : qa, *dst = (i & 1) ? qa : qb, *p = pool[i % P];
            k_real<REDB, GM><<<(C + 3) / 4, 256>>>(src, p, dst, flag, cnt, u, C, W);
        }
        if (rep == 1) CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s W=%d C=%d  %.2f us/launch  %.2f TB/s\n", name, W, C, ms * 1e3 / K, bytes / (ms * 1e-3 / K) / 1e12);
}

template <int W, bool RED, bool FLAGS>
static void run_work(const char *name, int C, int K, int P, double *qa, double *qb, std::vector<double *> &pool,
                     unsigned char *flag, long *cnt, double *u, hipEvent_t e0, hipEvent_t e1, double bytes)
{
    for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1) CK(hipEventRecord(e0));
        for (int i = 0; i < K; ++i) {
            double *src = (i & 1) ? qb : qa, *dst = (i & 1) ? qa : qb, *p = pool[i % P];
            k_work<W, RED, FLAGS><<<(C + 3) / 4, 256>>>(src, p, dst, flag, cnt, u, C);
        }
        if (rep == 1) CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s C=%d  %.2f us/launch  %.2f TB/s\n", name, C, ms * 1e3 / K, bytes / (ms * 1e-3 / K) / 1e12);
}

int main(int argc, char **argv)
{
    const int C = argc > 1 ? atoi(argv[1]) : 4096;
    const int P = 16, K = 200;
    const size_t n = (size_t)C * D;
    double *qa, *qb;
    std::vector<double *> pool(P);
    CK(hipMalloc(&qa, n * 8)); CK(hipMalloc(&qb, n * 8));
    // random operands: zero-filled buffers let the chip clock higher (DVFS) and
    // flatter the kernel
    std::vector<double> h(n);
    srand(1);
    const bool zeros = getenv("MEMBENCH_ZEROS") != nullptr;
    for (auto &x : h) x = zeros ? 0.0 : (rand() / (double)RAND_MAX - 0.5) * 3.4;
    CK(hipMemcpy(qa, h.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(qb, h.data(), n * 8, hipMemcpyHostToDevice));
    for (auto &p : pool) { CK(hipMalloc(&p, n * 8)); CK(hipMemcpy(p, h.data(), n * 8, hipMemcpyHostToDevice)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 24.0 * n;
    for (int variant = 0; variant < 4; ++variant) {
        const char *names[] = {"strided8 (current)", "coalesced16", "coalesced16+nt(p0)", "strided16 (2 chains/wave)"};
        for (int rep = 0; rep < 2; ++rep) {
            if (rep == 1) CK(hipEventRecord(e0));
            for (int i = 0; i < K; ++i) {
                double *src = (i & 1) ? qb : qa, *dst = (i & 1) ? qa : qb, *p = pool[i % P];
                if (variant == 0) k_strided8<<<(C + 3) / 4, 256>>>(src, p, dst, C);
                else if (variant == 1) k_coal16<false><<<(C + 3) / 4, 256>>>((d2v *)src, (d2v *)p, (d2v *)dst, C);
                else if (variant == 2) k_coal16<true><<<(C + 3) / 4, 256>>>((d2v *)src, (d2v *)p, (d2v *)dst, C);
                else k_strided16<<<(C / 2 + 3) / 4, 256>>>((d2v *)src, (d2v *)p, (d2v *)dst, C);
            }
            if (rep == 1) CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s C=%d  %.2f us/launch  %.2f TB/s\n", names[variant], C, ms * 1e3 / K, bytes / (ms * 1e-3 / K) / 1e12);
    }
    unsigned char *flag; long *cnt; double *u;
    CK(hipMalloc(&flag, C)); CK(hipMalloc(&cnt, C * 8)); CK(hipMalloc(&u, C * 8));
    CK(hipMemset(cnt, 0, C * 8)); CK(hipMemset(u, 0, C * 8));
    run_work<0, false, false>("work W=0", C, K, P, qa, qb, pool, flag, cnt, u, e0, e1, bytes);
    run_work<1, false, false>("work W=1", C, K, P, qa, qb, pool, flag, cnt, u, e0, e1, bytes);
    run_work<0, true, false>("work W=0 +reduce/exp", C, K, P, qa, qb, pool, flag, cnt, u, e0, e1, bytes);
    run_work<0, true, true>("work W=0 +reduce/exp+flags", C, K, P, qa, qb, pool, flag, cnt, u, e0, e1, bytes);
    run_work<1, true, true>("work W=1 +reduce/exp+flags", C, K, P, qa, qb, pool, flag, cnt, u, e0, e1, bytes);
    run_work<5, true, true>("work W=5 +reduce/exp+flags", C, K, P, qa, qb, pool, flag, cnt, u, e0, e1, bytes);
    run_work<20, true, true>("work W=20 +reduce/exp+flags", C, K, P, qa, qb, pool, flag, cnt, u, e0, e1, bytes);
    run_work<20, false, false>("work W=20", C, K, P, qa, qb, pool, flag, cnt, u, e0, e1, bytes);
    for (int W : {1, 20}) {
        for (int rep = 0; rep < 2; ++rep) {
            if (rep == 1) CK(hipEventRecord(e0));
            for (int i = 0; i < K; ++i) {
                double *src = (i & 1) ? qb : qa, *dst = (i & 1) ? qa : qb, *p = pool[i % P];
                int rc = binf_hmc_sample_gauss_f64(src, p, u, dst, flag, (int64_t *)cnt, nullptr, nullptr, 0.05, nullptr, C, D, W,
                                                   1.0, 0.0, 0, 1.05, 0.95, 0, nullptr);
                if (rc) { printf("rc=%d\n", rc); return 1; }
            }
            if (rep == 1) CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
        }
        { float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          printf("%-28s W=%d C=%d  %.2f us/launch  %.2f TB/s\n", "LIB kernel in this harness", W, C, ms * 1e3 / K, bytes / (ms * 1e-3 / K) / 1e12); }
        run_real<false, false>("real: red after only", W, C, K, P, qa, qb, pool, flag, cnt, u, e0, e1, bytes);
        run_real<true, false>("real: red before+after", W, C, K, P, qa, qb, pool, flag, cnt, u, e0, e1, bytes);
        run_real<true, true>("real: group-major", W, C, K, P, qa, qb, pool, flag, cnt, u, e0, e1, bytes);
    }
    return 0;
}

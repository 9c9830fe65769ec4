// Where does the time of the pair-distance force loop go when ONE 1024-thread
// workgroup owns a chain (the C5 per-GPU share: 256 chains = one workgroup per
// CU)?  The loop of quartered_force (binf_amd/csrc/pairdist.hip) with parts
// switched off, timed by events and by the wave's own cycle counters
// (s_memtime = shader clocks, s_memrealtime = 100 MHz), after a clock-settling
// spin.  VARIANT bits: 1 = no target-distance loads (y = const), 2 = no LDS reads
// (x_j from a register rotate), 4 = no rsq (w from a multiply), 8 = unroll 4.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/pairforce_probe.hip -o scripts/pairforce_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ inline double pair_weight(double d0, double d1, double d2, double y, bool norsq)
{
    const double s = (d0 * d0 + d1 * d1) + d2 * d2;
    double r = norsq ? s * 0.001 : __builtin_amdgcn_rsq(s);
    r = r * __builtin_fma(-0.5 * s * r, r, 1.5);
    return __builtin_fma(-y, r, 1.0);
}

template <int V>
__global__ void __launch_bounds__(1024) force(const double *x, const double *ymat, double *out,
                                              long long *clk, int n)
{
    extern __shared__ double sx[];
    const int c = blockIdx.x;
    const double *xc = x + (size_t)c * 3 * n;
    for (int k = threadIdx.x; k < 3 * n; k += 1024) sx[k] = xc[k];
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    const long long w0 = wall_clock64();
    const int qt = threadIdx.x & 3;
    const int i = threadIdx.x >> 2;
    const double x0 = sx[3 * i], x1 = sx[3 * i + 1], x2 = sx[3 * i + 2];
    const int per = (n + 3) >> 2;
    const int jb = qt * per;
    double f0 = 0.0, f1 = 0.0, f2 = 0.0;
    constexpr int U = (V & 8) ? 4 : 1;
    double r0 = x0 + 1.0, r1 = x1 - 1.0, r2 = x2 + 0.5;
    for (int t = 0; t < per; t += U) {
        double d[U][3], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jb + t + u;
            const int jc = j < n ? j : n - 1;
            if (V & 2) {
                r0 = r0 * 1.0000001; r1 = r1 * 0.9999999; r2 = r2 * 1.0000002;
                d[u][0] = x0 - r0; d[u][1] = x1 - r1; d[u][2] = x2 - r2;
            } else {
                d[u][0] = x0 - sx[3 * jc]; d[u][1] = x1 - sx[3 * jc + 1]; d[u][2] = x2 - sx[3 * jc + 2];
            }
            y[u] = (V & 1) ? 1.5 : ymat[(size_t)jc * n + i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jb + t + u;
            double w = pair_weight(d[u][0], d[u][1], d[u][2], y[u], (V & 4) != 0);
            w = (j != i && j < n) ? w : 0.0;
            f0 += w * d[u][0]; f1 += w * d[u][1]; f2 += w * d[u][2];
        }
    }
    f0 = f0 + __shfl_xor(f0, 1, 64); f0 = f0 + __shfl_xor(f0, 2, 64);
    f1 = f1 + __shfl_xor(f1, 1, 64); f1 = f1 + __shfl_xor(f1, 2, 64);
    f2 = f2 + __shfl_xor(f2, 1, 64); f2 = f2 + __shfl_xor(f2, 2, 64);
    const long long t1 = __builtin_readcyclecounter();
    const long long w1 = wall_clock64();
    if (qt == 0) {
        double *o = out + (size_t)c * 3 * n + 3 * i;
        o[0] = f0; o[1] = f1; o[2] = f2;
    }
    if (threadIdx.x == 0) { clk[2 * c] = t1 - t0; clk[2 * c + 1] = w1 - w0; }
}

// The loop with the NEXT round's loads (target distances and x_j) issued before the
// current round's arithmetic: two register buffers, ping-pong, scheduling barriers
// so that the compiler does not sink the loads to their first use again.
template <int U>
__global__ void __launch_bounds__(1024) force_pipe(const double *x, const double *ymat, double *out,
                                                   long long *clk, int n)
{
    extern __shared__ double sx[];
    const int c = blockIdx.x;
    const double *xc = x + (size_t)c * 3 * n;
    for (int k = threadIdx.x; k < 3 * n; k += 1024) sx[k] = xc[k];
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    const long long w0 = wall_clock64();
    const int qt = threadIdx.x & 3;
    const int i = threadIdx.x >> 2;
    const double x0 = sx[3 * i], x1 = sx[3 * i + 1], x2 = sx[3 * i + 2];
    const int per = (n + 3) >> 2;
    const int jb = qt * per;
    int je = jb + per; if (je > n) je = n;
    double f0 = 0.0, f1 = 0.0, f2 = 0.0;
    struct Buf { double y[U]; double xj[U][3]; };
    auto fetch = [&](Buf &b, int t) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jb + t + u;
            const int jc = j < n ? j : n - 1;
            b.y[u] = ymat[(size_t)jc * n + i];
            b.xj[u][0] = sx[3 * jc]; b.xj[u][1] = sx[3 * jc + 1]; b.xj[u][2] = sx[3 * jc + 2];
        }
    };
    auto compute = [&](const Buf &b, int t) {
        double d[U][3], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jb + t + u;
            d[u][0] = x0 - b.xj[u][0]; d[u][1] = x1 - b.xj[u][1]; d[u][2] = x2 - b.xj[u][2];
            const double wu = pair_weight(d[u][0], d[u][1], d[u][2], b.y[u], false);
            w[u] = (j != i && j < je) ? wu : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { f0 += w[u] * d[u][0]; f1 += w[u] * d[u][1]; f2 += w[u] * d[u][2]; }
    };
    Buf A, B;
    fetch(A, 0);
    for (int t = 0; t < per; t += 2 * U) {
        fetch(B, t + U);
        __builtin_amdgcn_sched_barrier(0);
        compute(A, t);
        __builtin_amdgcn_sched_barrier(0);
        fetch(A, t + 2 * U);
        __builtin_amdgcn_sched_barrier(0);
        compute(B, t + U);
        __builtin_amdgcn_sched_barrier(0);
    }
    f0 = f0 + __shfl_xor(f0, 1, 64); f0 = f0 + __shfl_xor(f0, 2, 64);
    f1 = f1 + __shfl_xor(f1, 1, 64); f1 = f1 + __shfl_xor(f1, 2, 64);
    f2 = f2 + __shfl_xor(f2, 1, 64); f2 = f2 + __shfl_xor(f2, 2, 64);
    const long long t1 = __builtin_readcyclecounter();
    const long long w1 = wall_clock64();
    if (qt == 0) {
        double *o = out + (size_t)c * 3 * n + 3 * i;
        o[0] = f0; o[1] = f1; o[2] = f2;
    }
    if (threadIdx.x == 0) { clk[2 * c] = t1 - t0; clk[2 * c + 1] = w1 - w0; }
}

__global__ void spin(double *o, int iters)
{
    double a = threadIdx.x * 1e-3, b = 1.0;
    for (int i = 0; i < iters; ++i) { a = a * 1.0000001 + b; b = b * 0.9999999 + a * 1e-9; }
    o[blockIdx.x * 256 + threadIdx.x] = a + b;
}

template <class K>
static void run_k(const char *name, K kern, int C, int n, const double *x, const double *ymat,
                  double *out, long long *clk);

template <int V>
static void run(const char *name, int C, int n, const double *x, const double *ymat, double *out,
                long long *clk)
{
    run_k(name, force<V>, C, n, x, ymat, out, clk);
}

template <class K>
static void run_k(const char *name, K kern, int C, int n, const double *x, const double *ymat,
                  double *out, long long *clk)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const size_t lds = 3 * n * sizeof(double);
    for (int w = 0; w < 20; ++w) kern<<<C, 1024, lds>>>(x, ymat, out, clk, n);
    hipEventRecord(a);
    const int reps = 200;
    for (int r = 0; r < reps; ++r) kern<<<C, 1024, lds>>>(x, ymat, out, clk, n);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(2 * C);
    hipMemcpy(h.data(), clk, sizeof(long long) * 2 * C, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int c = 0; c < C; ++c) { cyc += h[2 * c]; wall += h[2 * c + 1]; }
    cyc /= C; wall /= C;
    printf("%-34s C=%4d: %6.1f us per launch; loop %7.0f shader cycles = %5.1f us by the 100 MHz counter"
           " -> %.2f GHz\n", name, C, ms * 1e3 / reps, cyc, wall / 100.0, cyc / (wall / 100.0) * 1e-3);
}

int main(int argc, char **argv)
{
    const int n = 256, CMAX = 2048;
    double *x, *ymat, *out, *sp; long long *clk;
    hipMalloc(&x, sizeof(double) * CMAX * 3 * n);
    hipMalloc(&ymat, sizeof(double) * n * n);
    hipMalloc(&out, sizeof(double) * CMAX * 3 * n);
    hipMalloc(&sp, sizeof(double) * 1024 * 256);
    hipMalloc(&clk, sizeof(long long) * 2 * CMAX);
    std::vector<double> hx((size_t)CMAX * 3 * n), hy((size_t)n * n);
    srand(1);
    for (auto &v : hx) v = 4.0 * rand() / RAND_MAX - 2.0;
    for (auto &v : hy) v = 0.5 + 3.0 * rand() / RAND_MAX;
    hipMemcpy(x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(ymat, hy.data(), hy.size() * 8, hipMemcpyHostToDevice);
    for (int r = 0; r < 30; ++r) spin<<<1024, 256>>>(sp, 400000);      // settle the clocks
    hipDeviceSynchronize();
    for (int C : {16, 256}) {
        if (C == 16) {
            run<0>("as shipped before (1 pair / round)", 16, n, x, ymat, out, clk);
            run<8>("unroll 4", 16, n, x, ymat, out, clk);
            run<9>("unroll 4, no y loads", 16, n, x, ymat, out, clk);
            run<11>("unroll 4, no y loads, no LDS", 16, n, x, ymat, out, clk);
            run<15>("unroll 4, no loads, no LDS, no rsq", 16, n, x, ymat, out, clk);
            run<12>("unroll 4, no rsq", 16, n, x, ymat, out, clk);
            run<10>("unroll 4, no LDS", 16, n, x, ymat, out, clk);
        } else {
            run<0>("as shipped before (1 pair / round)", 256, n, x, ymat, out, clk);
            run<8>("unroll 4", 256, n, x, ymat, out, clk);
            run<9>("unroll 4, no y loads", 256, n, x, ymat, out, clk);
            run<11>("unroll 4, no y loads, no LDS", 256, n, x, ymat, out, clk);
            run<15>("unroll 4, no loads, no LDS, no rsq", 256, n, x, ymat, out, clk);
            run<12>("unroll 4, no rsq", 256, n, x, ymat, out, clk);
            run<10>("unroll 4, no LDS", 256, n, x, ymat, out, clk);
            run_k("pipelined, 2 pairs / round", force_pipe<2>, 256, n, x, ymat, out, clk);
            run_k("pipelined, 4 pairs / round", force_pipe<4>, 256, n, x, ymat, out, clk);
            run_k("pipelined, 8 pairs / round", force_pipe<8>, 256, n, x, ymat, out, clk);
        }
    }
    return 0;
}

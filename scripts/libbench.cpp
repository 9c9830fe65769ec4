// Development aid: time binf_hmc_sample_gauss_f64 from C++ (no Python in the loop).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../include/binf_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main(int argc, char **argv)
{
    const int C = argc > 1 ? atoi(argv[1]) : 4096;
    const int L = argc > 2 ? atoi(argv[2]) : 20;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    const int D = 1024, P = 16, K = 300;
    const size_t n = (size_t)C * D;
    std::vector<double> h(n);
    srand(1);
    for (auto &x : h) x = (rand() / (double)RAND_MAX - 0.5) * 3.4;
    double *qa, *qb, *u; unsigned char *acc; int64_t *nacc;
    std::vector<double *> pool(P);
    CK(hipMalloc(&qa, n * 8)); CK(hipMalloc(&qb, n * 8));
    CK(hipMemcpy(qa, h.data(), n * 8, hipMemcpyHostToDevice));
    for (auto &p : pool) { CK(hipMalloc(&p, n * 8)); CK(hipMemcpy(p, h.data(), n * 8, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&u, C * 8)); CK(hipMemset(u, 0, C * 8));
    CK(hipMalloc(&acc, C)); CK(hipMalloc(&nacc, C * 8)); CK(hipMemset(nacc, 0, C * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1) CK(hipEventRecord(e0));
        for (int i = 0; i < K; ++i) {
            double *src = (i & 1) ? qb : qa, *dst = (i & 1) ? qa : qb;
            int rc = binf_hmc_sample_gauss_f64(src, pool[i % P], u, dst, acc, nacc, nullptr, nullptr, 0.05,
                                               nullptr, C, D, L, 1.0, 0.0, 0, 1.05, 0.95, mode, nullptr);
            if (rc) { printf("rc=%d\n", rc); return 1; }
        }
        if (rep == 1) CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("lib C=%d L=%d mode=%d: %.2f us/launch, %.3e chain-steps/s\n", C, L, mode, ms * 1e3 / K,
           (double)C * L * K / (ms * 1e-3));
    return 0;
}

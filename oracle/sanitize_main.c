/* TEST INFRASTRUCTURE ONLY.  Driver for an AddressSanitizer / UndefinedBehaviorSanitizer build of
 * the C restatement (oracle_c.c): the checker every HIP kernel is held to is itself run under the
 * sanitizers this pool offers on the CPU (GPU ASan is not available).  Exact-size heap buffers, so
 * any read or write past an end is caught; ragged lengths around the pairwise-sum block / chunk
 * boundaries.  Built and run by tests/test_oracle.py::test_c_oracle_under_asan_and_ubsan. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

double oracle_np_sum(const double *a, int64_t n);
int oracle_hmc_sample_gauss(const double *q0, const double *p0, const double *u, double *q_out,
                            uint8_t *accepted, double *e_before, double *e_after, double *dt,
                            int64_t C, int64_t D, int32_t nsteps, double k, double x0, int32_t adapt,
                            double uprate, double downrate, int32_t nthreads);

int oracle_polyval(const double *xs, const double *coeffs, double *out, int64_t C, int64_t K, int64_t N);
int oracle_poly_gauss_logp(const double *coeffs, const double *xs, const double *ys, const double *precision,
                           double *out, double *chi2_out, int64_t C, int64_t K, int64_t N);

static double rnd(uint64_t *s)
{
    *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17;
    return (double)(*s >> 11) / 9007199254740992.0 * 2.0 - 1.0;
}

int main(void)
{
    uint64_t s = 0x9E3779B97F4A7C15ull;
    const int64_t lens[] = {0, 1, 7, 8, 9, 127, 128, 129, 255, 256, 1000, 1023, 1024, 1025, 8191, 8192,
                            8193, 16384, 20001};
    double check = 0.0;
    for (unsigned i = 0; i < sizeof(lens) / sizeof(lens[0]); ++i) {
        const int64_t n = lens[i];
        double *a = (double *)malloc(sizeof(double) * (n ? n : 1));
        for (int64_t j = 0; j < n; ++j) a[j] = rnd(&s);
        check += oracle_np_sum(a, n);
        free(a);
    }
    const int64_t shapes[][3] = {{1, 1, 1}, {3, 7, 2}, {5, 8, 3}, {4, 33, 20}, {2, 129, 4}, {3, 1024, 5},
                                 {2, 8193, 2}, {9, 258, 1}};
    for (unsigned i = 0; i < sizeof(shapes) / sizeof(shapes[0]); ++i) {
        const int64_t C = shapes[i][0], D = shapes[i][1];
        const int32_t L = (int32_t)shapes[i][2];
        double *q0 = (double *)malloc(sizeof(double) * C * D), *p0 = (double *)malloc(sizeof(double) * C * D);
        double *qo = (double *)malloc(sizeof(double) * C * D), *u = (double *)malloc(sizeof(double) * C);
        double *eb = (double *)malloc(sizeof(double) * C), *ea = (double *)malloc(sizeof(double) * C);
        double *dt = (double *)malloc(sizeof(double) * C);
        uint8_t *acc = (uint8_t *)malloc((size_t)C);
        for (int64_t j = 0; j < C * D; ++j) { q0[j] = rnd(&s); p0[j] = rnd(&s); }
        for (int64_t c = 0; c < C; ++c) { u[c] = 0.5 * (rnd(&s) + 1.0); dt[c] = 0.3 / sqrt((double)D); }
        for (int threads = 1; threads <= 2; ++threads) {
            const int rc = oracle_hmc_sample_gauss(q0, p0, u, qo, acc, eb, ea, dt, C, D, L, 2.5, 0.3,
                                                   threads == 2, 1.05, 0.95, threads);
            if (rc != 0) { fprintf(stderr, "rc=%d\n", rc); return 1; }
            for (int64_t c = 0; c < C; ++c) check += qo[c * D] + ea[c] + acc[c];
        }
        free(q0); free(p0); free(qo); free(u); free(eb); free(ea); free(dt); free(acc);
    }
    const int64_t pshapes[][3] = {{1, 1, 1}, {3, 4, 20}, {2, 33, 129}, {2, 9, 8193}, {4, 2, 7}};
    for (unsigned i = 0; i < sizeof(pshapes) / sizeof(pshapes[0]); ++i) {
        const int64_t C = pshapes[i][0], K = pshapes[i][1], N = pshapes[i][2];
        double *co = (double *)malloc(sizeof(double) * C * K), *xs = (double *)malloc(sizeof(double) * N);
        double *ys = (double *)malloc(sizeof(double) * N), *mock = (double *)malloc(sizeof(double) * C * N);
        double *pr = (double *)malloc(sizeof(double) * C), *lp = (double *)malloc(sizeof(double) * C);
        double *chi2 = (double *)malloc(sizeof(double) * C);
        for (int64_t j = 0; j < C * K; ++j) co[j] = rnd(&s);
        for (int64_t j = 0; j < N; ++j) { xs[j] = rnd(&s); ys[j] = rnd(&s); }
        for (int64_t c = 0; c < C; ++c) pr[c] = 1.5 + rnd(&s);
        if (oracle_polyval(xs, co, mock, C, K, N) != 0) return 4;
        if (oracle_poly_gauss_logp(co, xs, ys, pr, lp, (i & 1) ? chi2 : NULL, C, K, N) != 0) return 5;
        for (int64_t c = 0; c < C; ++c) check += mock[c * N + N - 1] + lp[c];
        free(co); free(xs); free(ys); free(mock); free(pr); free(lp); free(chi2);
    }
    if (oracle_polyval(NULL, NULL, NULL, 1, 0, 1) == 0) return 6;       /* K < 1 must be refused */
    if (oracle_hmc_sample_gauss(NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 1, 0, 1, 1.0, 0.0, 0, 1.05,
                                0.95, 1) == 0) return 2;               /* D < 1 must be refused */
    printf("sanitized oracle run ok (%.17g)\n", check);
    return isfinite(check) ? 0 : 3;
}

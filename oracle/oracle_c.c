/*
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Plain-C restatement of
 * the batched HMC transition, held bit-for-bit to oracle/ref_numpy.py.
 *
 * Parity status: held bit for bit to the numpy restatement AND to outputs of the
 * reference's own code (tests/golden/ref_leapfrog_*.npz, ref_sample_*.npz,
 * ref_adapt_timestep_*.npz -- oracle/gen_ref_leapfrog.py runs binf/samplers/hmc.py
 * itself, minus its one csb import): integrator, energies, bookkeeping, adaption.
 * Still **parity unpinned**: the definition of csb.numeric.exp (the clip bounds of
 * clipped_exp below), csb being absent from the reference tree.
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off: numpy rounds every
 * multiply and add separately, so no FMA contraction is allowed).
 *
 * Citations are relative to the reference root.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PW_BLOCKSIZE 128

/* numpy DOUBLE_pairwise_sum over f(a[i]); what np.sum() runs for the energy
 * reductions at binf/samplers/hmc.py:148,150 and binf/pdf/__init__.py:185. */
double oracle_pairwise_sum(const double *a, int64_t n)
{
    if (n < 8) {
        double res = -0.0;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= PW_BLOCKSIZE) {
        double r[8];
        int64_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) +
                     ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return oracle_pairwise_sum(a, n2) + oracle_pairwise_sum(a + n2, n - n2);
    }
}

/* np.sum(a): reduction seeded with the identity +0.0; numpy's buffered
 * reduction hands the inner loop consecutive chunks of NPY_BUFSIZE = 8192
 * elements and accumulates their pairwise sums one after the other. */
#define NPY_BUFSIZE 8192
double oracle_np_sum(const double *a, int64_t n)
{
    double res = 0.0;
    if (n <= 0) return res + oracle_pairwise_sum(a, 0);
    for (int64_t i = 0; i < n; i += NPY_BUFSIZE) {
        int64_t m = n - i < NPY_BUFSIZE ? n - i : NPY_BUFSIZE;
        res = res + oracle_pairwise_sum(a + i, m);
    }
    return res;
}

/* csb.numeric.exp (third party, unpinned): exp(clip(x, -308, 709)); NaN stays
 * NaN as numpy.clip propagates it. */
static double clipped_exp(double x)
{
    if (x < -308.0) x = -308.0;
    if (x > 709.0) x = 709.0;
    return exp(x);
}

/* -log_prob of the TestHO Gaussian: binf/pdf/__init__.py:181-185 with the
 * sign flip of binf/samplers/hmc.py:143.  tmp is D doubles of scratch. */
static double gauss_V(const double *x, int64_t D, double k, double x0,
                      double *tmp)
{
    for (int64_t i = 0; i < D; i++) {
        double d = x[i] - x0;
        tmp[i] = d * d;
    }
    double lp = (-0.5 * k) * oracle_np_sum(tmp, D);
    return -lp;
}

static double kinetic(const double *p, int64_t D, double *tmp)
{
    for (int64_t i = 0; i < D; i++) tmp[i] = p[i] * p[i];
    return 0.5 * oracle_np_sum(tmp, D);          /* hmc.py:148,150 */
}

/* gradient of the energy: binf/pdf/__init__.py:187-191 */
static inline double gauss_grad(double x, double k, double x0)
{
    return k * (x - x0);
}

/*
 * One HMCSampler.sample() (hmc.py:136-164) for each of C independent chains,
 * with the momentum draw p0[c] and the uniform u[c] supplied.
 *   dt        : per-chain timestep, length C, updated in place when adapt != 0
 *               (hmc.py:183-191: *uprate on accept, *downrate on reject)
 *   q_out[c]  : the returned sample (proposal if accepted, else q0[c])
 */
static inline double kick(double p, double dt, double g, int fused)
{
    return fused ? fma(-dt, g, p) : p - dt * g;            /* hmc.py:116,120,123 */
}

static inline double drift(double q, double p, double dt, int fused)
{
    return fused ? fma(p, dt, q) : q + p * dt;             /* hmc.py:119,122 */
}

static int hmc_sample_gauss(const double *q0, const double *p0, const double *u,
                            double *q_out, uint8_t *accepted, double *e_before,
                            double *e_after, double *dt, int64_t C, int64_t D,
                            int32_t nsteps, double k, double x0, int32_t adapt,
                            double uprate, double downrate, int32_t nthreads, int fused)
{
    if (C < 0 || D < 1 || nsteps < 1) return -1;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        double *q = (double *)malloc(sizeof(double) * D);
        double *p = (double *)malloc(sizeof(double) * D);
        double *tmp = (double *)malloc(sizeof(double) * D);
#pragma omp for schedule(static)
        for (int64_t c = 0; c < C; c++) {
            const double ts = dt[c];
            memcpy(q, q0 + c * D, sizeof(double) * D);
            memcpy(p, p0 + c * D, sizeof(double) * D);
            double Eb = gauss_V(q, D, k, x0, tmp) + kinetic(p, D, tmp);
            /* _leapfrog, hmc.py:116-123 */
            const double hts = 0.5 * ts;
            for (int64_t i = 0; i < D; i++)
                p[i] = kick(p[i], hts, gauss_grad(q[i], k, x0), fused);
            for (int32_t s = 0; s < nsteps - 1; s++) {
                for (int64_t i = 0; i < D; i++) q[i] = drift(q[i], p[i], ts, fused);
                for (int64_t i = 0; i < D; i++)
                    p[i] = kick(p[i], ts, gauss_grad(q[i], k, x0), fused);
            }
            for (int64_t i = 0; i < D; i++) q[i] = drift(q[i], p[i], ts, fused);
            for (int64_t i = 0; i < D; i++)
                p[i] = kick(p[i], hts, gauss_grad(q[i], k, x0), fused);
            double Ea = gauss_V(q, D, k, x0, tmp) + kinetic(p, D, tmp);
            int acc = u[c] < clipped_exp(-(Ea - Eb));      /* hmc.py:151 */
            e_before[c] = Eb;
            e_after[c] = Ea;
            accepted[c] = (uint8_t)acc;
            if (adapt) dt[c] = acc ? ts * uprate : ts * downrate;
            memcpy(q_out + c * D, acc ? q : q0 + c * D, sizeof(double) * D);
        }
        free(q);
        free(p);
        free(tmp);
    }
    return 0;
}

int oracle_hmc_sample_gauss(const double *q0, const double *p0, const double *u,
                            double *q_out, uint8_t *accepted, double *e_before,
                            double *e_after, double *dt, int64_t C, int64_t D,
                            int32_t nsteps, double k, double x0, int32_t adapt,
                            double uprate, double downrate, int32_t nthreads)
{
    return hmc_sample_gauss(q0, p0, u, q_out, accepted, e_before, e_after, dt, C, D, nsteps, k, x0,
                            adapt, uprate, downrate, nthreads, 0);
}

/*
 * The package's FMA mode (BINF_MODE_FMA, not a mode of the reference): the same
 * transition with each of the two leapfrog updates -- p -= dt * grad and q += p * dt --
 * as ONE correctly rounded fused multiply-add (C99 fma); the gradient k * (x - x0), the
 * half step 0.5 * dt, the energies and the accept test rounded as before.  Pins the
 * kernels' FMA arithmetic bit for bit (tests/test_gpu_hmc_gauss.py).
 */
int oracle_hmc_sample_gauss_fma(const double *q0, const double *p0, const double *u,
                                double *q_out, uint8_t *accepted, double *e_before,
                                double *e_after, double *dt, int64_t C, int64_t D,
                                int32_t nsteps, double k, double x0, int32_t adapt,
                                double uprate, double downrate, int32_t nthreads)
{
    return hmc_sample_gauss(q0, p0, u, q_out, accepted, e_before, e_after, dt, C, D, nsteps, k, x0,
                            adapt, uprate, downrate, nthreads, 1);
}

/* ---------------------------------------------------------------------------
 * The example's polynomial model (BASELINE C1 / C3 / C4), the parts whose bits are numpy's.
 *
 * ForwardModel._evaluate (binf/example/likelihood.py:24-26) with
 * numpy.polynomial.polynomial.polyval as the `polynomial` (example_script.py:21): Horner from the
 * highest coefficient, c0 = c[-1] + x*0;  c0 = c[-i] + c0*x  (oracle/ref_numpy.py:225-233).
 * out[c*N + n] = polyval(xs[n], coeffs[c*K ...]).
 */
int oracle_polyval(const double *xs, const double *coeffs, double *out, int64_t C, int64_t K, int64_t N)
{
    if (C < 0 || K < 1 || N < 0) return -1;
    for (int64_t c = 0; c < C; c++) {
        const double *co = coeffs + c * K;
        for (int64_t n = 0; n < N; n++) {
            const double x = xs[n];
            double c0 = co[K - 1] + x * 0;
            for (int64_t i = 2; i <= K; i++) c0 = co[K - i] + c0 * x;
            out[c * N + n] = c0;
        }
    }
    return 0;
}

/*
 * Likelihood._evaluate_log_prob (binf/pdf/likelihoods.py:141-146) for that forward model and
 * GaussianErrorModel._evaluate_log_prob (binf/example/likelihood.py:54-57):
 *   -0.5 * np.sum((mock - ys)**2) * precision + len(ys) * 0.5 * np.log(precision)
 * np.sum in numpy's order (oracle_np_sum); `log` is this host's libm -- numpy's own log differs
 * between CPUs, so only precisions whose log is exact (1, 2, 4 ...) make the LAST term a fixed
 * bit pattern; chi^2 and its scaling always are.  chi2_out (may be NULL) receives np.sum(...).
 */
int oracle_poly_gauss_logp(const double *coeffs, const double *xs, const double *ys, const double *precision,
                           double *out, double *chi2_out, int64_t C, int64_t K, int64_t N)
{
    if (C < 0 || K < 1 || N < 1) return -1;
    double *mock = (double *)malloc(sizeof(double) * (size_t)N);
    if (!mock) return -2;
    for (int64_t c = 0; c < C; c++) {
        oracle_polyval(xs, coeffs + c * K, mock, 1, K, N);
        for (int64_t n = 0; n < N; n++) {
            const double d = mock[n] - ys[n];
            mock[n] = d * d;
        }
        const double chi2 = oracle_np_sum(mock, N);
        const double p = precision[c];
        const double logZ = (double)N * 0.5 * log(p);
        out[c] = -0.5 * chi2 * p + logZ;
        if (chi2_out) chi2_out[c] = chi2;
    }
    free(mock);
    return 0;
}

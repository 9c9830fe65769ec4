/*
 * binf_hip.h -- C ABI of libbinf_hip.so, the MI355X (gfx950) engine behind
 * binf's HMC hot path.  Plain C types only; every pointer marked "device" is
 * HBM memory owned by the CALLER (e.g. a PyTorch-ROCm tensor's data_ptr()).
 *
 * Conventions (all functions)
 *   - return 0 on success, <0 for an argument error (BINF_E_*), >0 = hipError_t.
 *     Nothing is thrown across the ABI; binf_last_error() gives the text.
 *   - stateless and re-entrant; work is enqueued on `stream` (a hipStream_t,
 *     NULL = the default stream) and the call returns without synchronising.
 *   - the library allocates nothing persistent, keeps no pointer past return
 *     and never frees caller memory.  Safe inside hipGraph stream capture.
 *   - layout: chain-major row-major fp64, x[c*D + i] for chain c, dimension i.
 *
 * The reference (simeoncarstens/binf) has no native code and no FFI; each
 * entry point below names the reference Python it replaces (paths relative to
 * the reference root).
 */
#ifndef BINF_HIP_H
#define BINF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: every entry point that GENERATES draws takes the global index of its first
 *    chain / element (chain_offset, elem_offset) so that a sharded run reproduces
 *    the unsharded one bit for bit; new entry points for the RWMC subsampler.
 *    (1 -> 2 also covers the contract changes made late in ABI 1: the polynomial
 *    gradient's workspace is mandatory, binf_hmc_sample_poly_f64 accepts N <= 1024.)
 * 3: binf_hmc_sample_poly_f64 spreads a chain's data over a lane group for EVERY
 *    N <= 1024 (the force's summation order for N <= 128 changed with it; one lane
 *    per chain is now the opt-in BINF_MODE_LANE_PER_CHAIN); binf_gibbs_poly_sample_n_f64,
 *    binf_jacobian_contract_f64, binf_sum_terms_f64, binf_poly_leapfrog_f64,
 *    binf_poly_gauss_logp_memo_f64, binf_pairdist_gauss_logp_memo_f64,
 *    binf_hmc_sample_n_gauss_big_f64 / _rng_f64.
 * 4: binf_pairdist_packed_targets_bytes, binf_pairdist_pack_targets_f64,
 *    binf_pairdist_gauss_grad_packed_f64, binf_pairdist_leapfrog_packed_f64,
 *    binf_pairdist_hmc_energy_f64, binf_rng_normal_zig_uniform_f64; the chi^2
 *    memos of binf_poly_gauss_logp_memo_f64 / binf_pairdist_gauss_logp_memo_f64 hold two
 *    entries per chain (their buffers are twice the ABI 3 size).
 * 5: binf_sum_terms_bcast_f64 (terms that are ONE device double, broadcast: a 0-dim
 *    tensor never has to be read back to the host); binf_hmc_sample_n_gauss_rng_f64 /
 *    binf_hmc_gauss_rng_draws_f64 take one stream position per TRANSITION (offset + i),
 *    as the long-chain entry points always did.
 * 6: optional workspace for the pair-distance log-prob / energy entry points (chi^2 by chunks with
 *    few chains: binf_pairdist_chi2_workspace_bytes); packed targets and the ring kernels for 257..1024 beads (binf_pairdist_packed_targets_bytes
 *    is no longer 0 there), the `_packed_` entry points take an optional workspace
 *    (binf_pairdist_tiles_workspace_bytes: a wave per tile when there are few chains); binf_predictive_density_f64 / _workspace_bytes (the consumer side of the sample store: the
 *    posterior-predictive density over a grid of points in one launch). */
#define BINF_ABI_VERSION 6

#define BINF_E_ARG        (-1)  /* null pointer / negative size / bad flag    */
#define BINF_E_UNSUPPORTED (-2) /* shape outside what the kernels cover       */
#define BINF_E_ALIAS      (-3)  /* illegal partial overlap of buffers         */

/* arithmetic mode of the leapfrog update */
#define BINF_MODE_EXACT 0  /* every multiply and add rounded separately, numpy
                              pairwise-sum order: bit-identical to the CPU
                              restatement of the reference's numpy path      */
#define BINF_MODE_FMA   1  /* p-=dt*g and q+=p*dt contracted to one FMA each:
                              within 1e-10 relative of EXACT, not bit-equal  */
/* flag, OR-ed into `mode` of binf_hmc_sample_poly_f64 only: one LANE per chain
 * (N <= 128) instead of a lane group -- the faster layout from ~10^5 chains up.
 * Energies are the same bits; the force is summed in data order instead of
 * per-lane partial sums + butterfly, so trajectories differ at rounding level. */
#define BINF_MODE_LANE_PER_CHAIN 16

int32_t binf_abi_version(void);

/* Copies the calling thread's last error text (NUL-terminated, truncated to
 * n) into buf; returns its full length. */
int32_t binf_last_error(char *buf, size_t n);

/* ------------------------------------------------------------------------
 * Fused HMC transition on an isotropic Gaussian  (BASELINE config C2)
 *
 * Replaces, for C independent chains in one launch:
 *   HMCSampler.sample()          binf/samplers/hmc.py:136-164
 *   HMCSampler._leapfrog()       binf/samplers/hmc.py:92-125
 *   HMCSampler._adapt_timestep() binf/samplers/hmc.py:183-191
 *   TestHO log_prob / gradient   binf/pdf/__init__.py:181-191
 *       log p(x) = -0.5*k*sum((x-x0)**2),  gradient = k*(x-x0)  [of -log p]
 *
 * Per chain c:   p = p0[c];  E_b = V(q0)+0.5*sum(p**2);  leapfrog nsteps;
 *   E_a = V(q)+0.5*sum(p**2);  acc = u[c] < exp(clip(-(E_a-E_b),-308,709));
 *   q_out[c] = acc ? q : q0[c].
 *
 *   q0, p0      device, [C*D]   start positions; momentum draws (the
 *                               np.random.normal(size=D) of hmc.py:146)
 *   u           device, [C]     uniform draws (hmc.py:151)
 *   q_out       device, [C*D]   returned samples.  May be EXACTLY q0 (state
 *                               updated in place) or disjoint from q0/p0.
 *   accepted    device, [C]     1/0
 *   n_accepted  device, [C] or NULL  per-chain acceptance counter, += 1 on
 *                               accept (hmc.py:161)
 *   e_before/e_after device,[C] energies of hmc.py:148,150 (may be NULL)
 *   timestep    host scalar     used when dt_chain == NULL
 *   dt_chain    device, [C] or NULL  per-chain timestep; when adapt != 0 it is
 *                               updated in place: *uprate on accept,
 *                               *downrate on reject (hmc.py:188-191).
 *                               adapt != 0 requires dt_chain != NULL.
 *   mode        BINF_MODE_EXACT | BINF_MODE_FMA
 * Supported: 1 <= D <= 8192 whose numpy pairwise-sum tree has height <= 6
 * (every D <= 7400 and all multiples of 64 up to 8192; one wave per chain up to
 * height 3, i.e. D <= ~1024, then 2 / 4 / 8 waves per chain); nsteps >= 1.
 * ---------------------------------------------------------------------- */
int32_t binf_hmc_sample_gauss_f64(const double *q0, const double *p0,
                                  const double *u, double *q_out,
                                  uint8_t *accepted, int64_t *n_accepted,
                                  double *e_before,
                                  double *e_after, double timestep,
                                  double *dt_chain, int64_t C, int64_t D,
                                  int32_t nsteps, double k, double x0,
                                  int32_t adapt, double uprate, double downrate,
                                  int32_t mode, void *stream);

/* ------------------------------------------------------------------------
 * Persistent variant: n consecutive transitions of every chain in ONE launch
 * -- the `for i in range(n): sampler.sample()` loop of example_script.py:33-34
 * for the same Gaussian.  Bit-identical to n calls of
 * binf_hmc_sample_gauss_f64 (same arithmetic, same order); the state stays in
 * registers between transitions, so HBM sees only the momentum draws in and
 * the recorded states out.
 *
 *   p0        device, [n*C*D]   draw s at p0 + s*C*D
 *   u         device, [n*C]
 *   q_out     device, [C*D]     state after the n-th transition (may be == q0)
 *   samples   device, [(n/thin)*C*D] or NULL: the state after transitions
 *             thin, 2*thin, ... (thin >= 1; thinning as example_script.py:41)
 *   accepted, e_before, e_after   device, [n*C] or NULL (per transition)
 *   n_accepted device, [C] or NULL: += number of accepted transitions
 *   n_adapt   the first n_adapt transitions adapt dt_chain (hmc.py:156-157:
 *             a caller with `counter` samples drawn and adaption limit `lim`
 *             passes max(0, min(n, lim - 1 - counter)))
 * ---------------------------------------------------------------------- */
int32_t binf_hmc_sample_n_gauss_f64(const double *q0, const double *p0,
                                    const double *u, double *q_out,
                                    double *samples, uint8_t *accepted,
                                    int64_t *n_accepted, double *e_before,
                                    double *e_after, double timestep,
                                    double *dt_chain, int64_t C, int64_t D,
                                    int32_t nsteps, int32_t n, int32_t thin,
                                    double k, double x0, int32_t n_adapt,
                                    double uprate, double downrate,
                                    int32_t mode, void *stream);

/* Waves per chain binf_hmc_sample_[n_]gauss_f64 uses for a [C x D] batch: 1, or
 * 2 / 4 for few chains of D = 768 / 1024 (a chain spread over several waves), or
 * 2 / 4 / 8 for 1024 < D <= 8192; 0 if the shape is not covered.  Host function. */
int32_t binf_hmc_gauss_waves_per_chain(int64_t C, int64_t D);

/* ------------------------------------------------------------------------
 * One transition for chains of ANY length -- what binf_hmc_sample_gauss_f64
 * does not cover (D > 8192; lengths whose pairwise tree is deeper than 6).
 * numpy sums long vectors 8192 elements at a time, so the trajectory runs per
 * (chain, 8192-chunk) with the state in registers, the chunk sums are chained
 * per chain, and rejected chains are restored from q0: three launches and
 * 24 D bytes per chain instead of ~3 L launches and 32 D bytes per leapfrog
 * step of the per-step tier; same bits.  Arguments as
 * binf_hmc_sample_gauss_f64, plus
 *   workspace  device scratch of binf_hmc_sample_gauss_big_workspace_bytes(C, D)
 *              bytes (caller-owned; too small or NULL -> BINF_E_ARG)
 * q_out must not overlap q0 or p0 (BINF_E_ALIAS).
 * ---------------------------------------------------------------------- */
int64_t binf_hmc_sample_gauss_big_workspace_bytes(int64_t C, int64_t D);
int32_t binf_hmc_sample_gauss_big_f64(const double *q0, const double *p0,
                                      const double *u, double *q_out,
                                      uint8_t *accepted, int64_t *n_accepted,
                                      double *e_before, double *e_after,
                                      double timestep, double *dt_chain,
                                      int64_t C, int64_t D, int32_t nsteps,
                                      double k, double x0, int32_t adapt,
                                      double uprate, double downrate,
                                      int32_t mode, void *workspace,
                                      int64_t workspace_bytes, void *stream);

/* n consecutive long-chain transitions from one call: the loop
 * `for i in range(n): sampler.sample()` (example_script.py:33-34) for chains of
 * any length.  Transition s writes its proposal straight into its record slot
 * samples[(s + 1) / thin - 1] (or into q_out / a scratch state when it is not
 * recorded), rejected chains are restored from the state the transition read,
 * and transition s + 1 reads what s wrote: 3 launches and 24 D bytes per chain
 * and transition, recorded or not (a loop of single calls copies every recorded
 * state once more).  BIT-IDENTICAL to n calls of binf_hmc_sample_gauss_big_f64
 * (the same kernels).  p0 [n x C x D], u [n x C], accepted / e_before / e_after
 * [n x C], samples [n / thin x C x D] or NULL; the first n_adapt transitions adapt
 * dt_chain.  workspace: binf_hmc_sample_n_gauss_big_workspace_bytes(C, D) bytes
 * (chunk sums + one [C x D] scratch state).  q_out holds the state after
 * transition n.  The _rng form generates the draws in the kernels, transition s
 * under (seed, offset + s): exactly the draws of n calls of
 * binf_hmc_sample_gauss_big_rng_f64 with offsets offset, offset + 1, ...
 * (No persistent kernel: the trajectory kernel moves its 24 D bytes as fast per
 * byte as the persistent kernel of binf_hmc_sample_n_gauss_f64 moves its 16 D;
 * chains longer than a workgroup's registers stream through HBM either way.) */
int64_t binf_hmc_sample_n_gauss_big_workspace_bytes(int64_t C, int64_t D);
int32_t binf_hmc_sample_n_gauss_big_f64(const double *q0, const double *p0,
                                        const double *u, double *q_out, double *samples,
                                        uint8_t *accepted, int64_t *n_accepted,
                                        double *e_before, double *e_after,
                                        double timestep, double *dt_chain, int64_t C,
                                        int64_t D, int32_t nsteps, int32_t n,
                                        int32_t thin, double k, double x0,
                                        int32_t n_adapt, double uprate, double downrate,
                                        int32_t mode, void *workspace,
                                        int64_t workspace_bytes, void *stream);
int32_t binf_hmc_sample_n_gauss_big_rng_f64(const double *q0, double *q_out,
                                            double *samples, uint8_t *accepted,
                                            int64_t *n_accepted, double *e_before,
                                            double *e_after, double timestep,
                                            double *dt_chain, int64_t C, int64_t D,
                                            int32_t nsteps, int32_t n, int32_t thin,
                                            double k, double x0, int32_t n_adapt,
                                            double uprate, double downrate, int32_t mode,
                                            uint64_t seed, uint64_t offset,
                                            int64_t chain_offset, void *workspace,
                                            int64_t workspace_bytes, void *stream);

/* The long-chain transition with its draws generated inside the kernels (one
 * xoshiro128++ stream per lane and (chain, 8192-chunk) for the momentum, one per
 * chain for the acceptance draw; keyed by (seed, offset), a caller advances
 * offset by one per call): binf_hmc_sample_gauss_big_f64 without p0 / u.
 * chain_offset (>= 0) is the GLOBAL index of chain 0 of this call: the streams are
 * keyed by global chain, so a rank that owns chains [s, s + C) of a larger run
 * passes chain_offset = s and draws exactly what the unsharded call draws for them.
 * binf_hmc_gauss_big_rng_draws_f64 writes those draws out instead (p0_out [C*D],
 * u_out [C]); feeding them to binf_hmc_sample_gauss_big_f64 reproduces the fused
 * call bit for bit. */
int32_t binf_hmc_sample_gauss_big_rng_f64(const double *q0, double *q_out,
                                          uint8_t *accepted, int64_t *n_accepted,
                                          double *e_before, double *e_after,
                                          double timestep, double *dt_chain,
                                          int64_t C, int64_t D, int32_t nsteps,
                                          double k, double x0, int32_t adapt,
                                          double uprate, double downrate,
                                          int32_t mode, uint64_t seed,
                                          uint64_t offset, int64_t chain_offset,
                                          void *workspace,
                                          int64_t workspace_bytes, void *stream);
int32_t binf_hmc_gauss_big_rng_draws_f64(double *p0_out, double *u_out, int64_t C,
                                         int64_t D, uint64_t seed, uint64_t offset,
                                         int64_t chain_offset, void *stream);

/* ------------------------------------------------------------------------
 * The same n transitions with the random draws generated INSIDE the kernel:
 * HMCSampler.sample() as the reference defines it, np.random.normal(size=
 * q.shape) ... np.random.uniform() (binf/samplers/hmc.py:146,151), with no
 * momentum buffer in HBM.  Per-lane xoshiro128++ streams seeded from the Philox
 * block (lane's element set, offset) under `seed`, normals by a 1024-layer
 * ziggurat; chain c of the call draws the stream of GLOBAL chain chain_offset + c:
 * deterministic in (seed, offset, global chain, D), independent of the launch
 * geometry, of the batch size C and of how a run is sharded over GPUs (a rank that
 * owns chains [s, s + C) passes chain_offset = s); NOT numpy's MT19937 stream
 * (parity runs inject host draws through binf_hmc_sample_n_gauss_f64).  Transition
 * i of a launch draws from stream position offset + i -- what a single-transition
 * launch at that position draws, so n transitions in one launch equal n launches of
 * one (ABI 5; before, a launch ran on from ONE position) -- and a caller advances
 * `offset` by n per launch.
 * Arguments as binf_hmc_sample_n_gauss_f64 without p0 / u.
 * Supported: the shapes of binf_hmc_sample_n_gauss_f64 (D <= 8192, pairwise
 * tree height <= 6); otherwise BINF_E_UNSUPPORTED (use the stand-alone
 * generators below + binf_hmc_sample_gauss_big_f64).
 * ---------------------------------------------------------------------- */
int32_t binf_hmc_sample_n_gauss_rng_f64(const double *q0, double *q_out,
                                        double *samples, uint8_t *accepted,
                                        int64_t *n_accepted, double *e_before,
                                        double *e_after, double timestep,
                                        double *dt_chain, int64_t C, int64_t D,
                                        int32_t nsteps, int32_t n, int32_t thin,
                                        double k, double x0, int32_t n_adapt,
                                        double uprate, double downrate,
                                        int32_t mode, uint64_t seed,
                                        uint64_t offset, int64_t chain_offset,
                                        void *stream);

/* The draws binf_hmc_sample_n_gauss_rng_f64 consumes for (seed, offset, chain_offset, C, D, n),
 * written out instead of used: p0_out [n*C*D], u_out [n*C].  Feeding them to
 * binf_hmc_sample_n_gauss_f64 reproduces the fused call bit for bit -- the
 * handle by which the fused generator is tested and its stream inspected. */
int32_t binf_hmc_gauss_rng_draws_f64(double *p0_out, double *u_out, int64_t C,
                                     int64_t D, int32_t n, uint64_t seed,
                                     uint64_t offset, int64_t chain_offset,
                                     void *stream);

/* ------------------------------------------------------------------------
 * Generic per-step tier: the pieces of HMCSampler.sample() as separate
 * chain-batched launches, for posteriors whose gradient comes from other code
 * (any plug-in with log_prob / gradient, reference binf/samplers/hmc.py:114,143).
 * ---------------------------------------------------------------------- */

/* Row reductions in numpy's pairwise order, bit-identical to np.sum on each
 * row:  out[c] = scale * np.sum(f(x[c,:])).
 *   op = BINF_ROW_SUM          f(x) = x
 *        BINF_ROW_SUMSQ        f(x) = x*x            (0.5*np.sum(p**2): scale=0.5,
 *                                                     hmc.py:148,150)
 *        BINF_ROW_SUMSQ_SHIFT  f(x) = (x-shift)**2   (binf/pdf/__init__.py:185)
 * (rows longer than numpy's 8192-element buffer are accumulated chunk by chunk,
 * as numpy does).  x device [C*D], out device [C]; any D < 2^31. */
#define BINF_ROW_SUM 0
#define BINF_ROW_SUMSQ 1
#define BINF_ROW_SUMSQ_SHIFT 2
int32_t binf_row_sum_f64(const double *x, double *out, int64_t C, int64_t D,
                         int32_t op, double shift, double scale, void *stream);

/* HMCSampler.sample()'s energy (binf/samplers/hmc.py:143,148,150)
 *   out[c] = -log_prob[c] + 0.5 * np.sum(p[c,:]**2)
 * in one launch (the kinetic row sum in numpy's order, the subtraction as its
 * epilogue; bit-identical to negating, summing and adding separately).
 * p device [C*D], log_prob / out device [C] (out may alias log_prob). */
int32_t binf_hmc_energy_f64(const double *p, const double *log_prob, double *out,
                            int64_t C, int64_t D, void *stream);

/* p[c,:] -= (half ? 0.5*dt : dt) * grad[c,:]     hmc.py:116,120,123
 * dt = dt_chain[c] if dt_chain != NULL else timestep. */
int32_t binf_leapfrog_kick_f64(double *p, const double *grad, double timestep,
                               const double *dt_chain, int32_t half, int64_t C,
                               int64_t D, int32_t mode, void *stream);

/* q[c,:] += p[c,:] * dt                           hmc.py:119,122 */
int32_t binf_leapfrog_drift_f64(double *q, const double *p, double timestep,
                                const double *dt_chain, int64_t C, int64_t D,
                                int32_t mode, void *stream);

/* p[c,:] -= dt * grad[c,:];  q[c,:] += p[c,:] * dt    hmc.py:120 then :119 / :122
 * (one interior leapfrog step after its gradient call; the same roundings as
 * binf_leapfrog_kick_f64 followed by binf_leapfrog_drift_f64, one pass). */
int32_t binf_leapfrog_kick_drift_f64(double *q, double *p, const double *grad,
                                     double timestep, const double *dt_chain,
                                     int64_t C, int64_t D, int32_t mode,
                                     void *stream);

/* out = k * (x - x0): energy gradient of TestHO, binf/pdf/__init__.py:187-191 */
int32_t binf_gauss_grad_f64(const double *x, double *out, double k, double x0,
                            int64_t C, int64_t D, void *stream);

/* out[i] = exp(clip(x[i], -308, 709)): csb.numeric.exp as the reference uses it
 * in the accept test (binf/samplers/hmc.py:10,151) and in
 * AbstractBinfPDF._evaluate (binf/pdf/__init__.py:10,89).  The SAME device
 * function decides every accept test of this library. */
int32_t binf_clipped_exp_f64(const double *x, double *out, int64_t n, void *stream);

/* acc[c] = u[c] < exp(clip(-(e_after[c]-e_before[c]), -308, 709));
 * q_out[c,:] = acc ? q_prop[c,:] : q_old[c,:];  optional step-size adaption of
 * dt_chain as in binf_hmc_sample_gauss_f64.       hmc.py:151-164,188-191
 * n_accepted (device [C] or NULL) += 1 on accept.
 * q_out may be exactly q_prop or exactly q_old. */
int32_t binf_accept_select_f64(const double *q_prop, const double *q_old,
                               const double *e_before, const double *e_after,
                               const double *u, double *q_out, uint8_t *accepted,
                               int64_t *n_accepted, double *dt_chain,
                               int32_t adapt, double uprate,
                               double downrate, int64_t C, int64_t D,
                               void *stream);

/* out[c] = scale * np.sum((x[c,:] - y[:])**2 / w[:]), numpy order; y, w device
 * [D]; w == NULL means no division.  The chi^2 of GaussianErrorModel
 * (binf/example/likelihood.py:57) and the GaussianPrior exponent
 * (binf/example/priors.py:54). */
int32_t binf_row_sumsq_diff_f64(const double *x, const double *y, const double *w,
                                double *out, int64_t C, int64_t D, double scale,
                                void *stream);

/* ------------------------------------------------------------------------
 * Polynomial forward model + Gaussian error model (the reference's example
 * application, BASELINE configs C1/C3/C4).  K coefficients (<= 64), N data.
 *   precision / precision_chain: host scalar, or device [C] per-chain values
 *   (non-NULL wins) -- the Gibbs state's 'precision' variable.
 * ---------------------------------------------------------------------- */

/* out[c,n] = polyval(xs[n], coeffs[c,:]), Horner in numpy's order (bit-exact).
 * ForwardModel._evaluate, binf/example/likelihood.py:24-26.  C <= 65535. */
int32_t binf_poly_forward_f64(const double *coeffs, const double *xs, double *out,
                              int64_t C, int64_t K, int64_t N, void *stream);

/* out[c,n] = (mock[c,n] - ys[n]) * precision_c.
 * GaussianErrorModel._evaluate_gradient, likelihood.py:59-61.  C <= 65535. */
int32_t binf_gauss_err_grad_f64(const double *mock, const double *ys,
                                double precision, const double *precision_chain,
                                double *out, int64_t C, int64_t N, void *stream);

/* out[c] = -0.5*np.sum((mock[c,:]-ys)**2)*precision_c + N*0.5*log(precision_c).
 * GaussianErrorModel._evaluate_log_prob, likelihood.py:54-57. */
int32_t binf_gauss_err_logp_f64(const double *mock, const double *ys,
                                double precision, const double *precision_chain,
                                double *out, int64_t C, int64_t N, void *stream);

/* Likelihood._evaluate_log_prob (binf/pdf/likelihoods.py:141-146) for the
 * polynomial + Gaussian pair, fused: the mock data is never materialised.
 * chi^2 is bit-identical to the numpy path; log() is the device's. */
int32_t binf_poly_gauss_logp_f64(const double *coeffs, const double *xs,
                                 const double *ys, double precision,
                                 const double *precision_chain, double *out,
                                 int64_t C, int64_t K, int64_t N, void *stream);

/* The same log-prob with a per-chain MEMO of chi^2 = np.sum((polyval - ys)**2), the
 * expensive part (a Horner pass over all N data points per chain), which depends on the
 * coefficients alone.  The memo has TWO entries per chain (HMCSampler.sample() evaluates
 * the state and the proposal, hmc.py:148,150, and the next call's state is one of the two
 * whichever way the acceptance test went): memo_coeffs [2 x C x K] / memo_chi2 [2 x C]
 * (caller-owned, device; fill both with NaN before the first use) hold the coefficients
 * a stored chi^2 belongs to; memo_state [2 x C] bytes (zero before the first use) is the
 * memo's bookkeeping: [0..C) 1 = this call reused a stored chi^2 for the chain, [C..2C)
 * the entry it used or refilled.  A chain whose coefficients equal one entry's BIT FOR BIT
 * takes that entry's chi^2; every other chain is summed and replaces the entry it did not
 * use last.  The check runs on the device, chain by chain -- no tensor identities,
 * versions or host synchronisation are involved, so the results are those of
 * binf_poly_gauss_logp_f64 bit for bit whatever happened to the buffers in between.
 * In a Gibbs sweep (binf/samplers/gibbs.py:146-149) the log-prob is asked for the
 * proposal (hmc.py:150), again for the sweep's new coefficients at precision = 1
 * (binf/example/samplers.py:34-41) and once more as the next sweep's E_before
 * (hmc.py:148): one Horner pass instead of three, accepted or not. */
int32_t binf_poly_gauss_logp_memo_f64(const double *coeffs, const double *xs,
                                      const double *ys, double precision,
                                      const double *precision_chain, double *out,
                                      double *memo_coeffs, double *memo_chi2,
                                      uint8_t *memo_state, int64_t C, int64_t K, int64_t N,
                                      void *stream);

/* Likelihood._evaluate_gradient (binf/pdf/likelihoods.py:148-155) for the same
 * pair: out[c,:] = J . ((mock_c - ys) * precision_c) with the Jacobian
 * J = design [K x N] (design[i][n] = xs[n]**i, likelihood.py:28-30), as two
 * chained f64 MFMA products per tile.  workspace: device scratch of
 * binf_poly_gauss_grad_workspace_bytes(C,K,N) bytes (caller-owned; 0 bytes
 * needed -> may be NULL; NULL or too small otherwise -> BINF_E_ARG, the
 * summation order is never changed silently).  The data range is summed in a
 * number of pieces that follows from (C, N), partial sums joined in a fixed
 * order: a call is deterministic, but the same chain evaluated in batches of
 * different size (one GPU vs a shard) can differ at rounding level -- inside
 * the 1e-10 bar the contraction is held to (the reference's BLAS order is not
 * reproducible either). */
int64_t binf_poly_gauss_grad_workspace_bytes(int64_t C, int64_t K, int64_t N);
int32_t binf_poly_gauss_grad_f64(const double *coeffs, const double *design,
                                 const double *ys, double precision,
                                 const double *precision_chain, double *out,
                                 void *workspace, int64_t workspace_bytes,
                                 int64_t C, int64_t K, int64_t N, void *stream);

/* HMCSampler._leapfrog (binf/samplers/hmc.py:92-125) under the force of the
 * polynomial + Gaussian likelihood alone (the example's conditional posterior of
 * the coefficients: its priors are not differentiable, posteriors.py:183), IN
 * PLACE on q = coefficients [C x K] and p [C x K]: per gradient call one launch of
 * the MFMA gradient kernel of binf_poly_gauss_grad_f64 (partial sums over the
 * data splits) and ONE launch that adds the partial sums in split order and
 * applies the kick (hmc.py:116,120,123) and the drift that follows it (:119,122)
 * -- 2 (nsteps + 1) launches from one call instead of ~3 per step through the
 * class stack.  BIT-IDENTICAL to the per-step sequence binf_poly_gauss_grad_f64,
 * binf_leapfrog_kick_f64 (half), binf_leapfrog_drift_f64,
 * binf_leapfrog_kick_drift_f64 ... on the same batch.  (Combining a chain tile
 * inside the gradient launch -- last workgroup, agent-scope release / acquire --
 * was built and measured: 68-130 us per step slower, a tile's 16 x 34 KB of
 * partial sums are too much for one workgroup to read.)  workspace: caller-owned
 * device scratch of binf_poly_leapfrog_workspace_bytes(C, K, N) bytes
 * (mandatory).  timestep / dt_chain, mode as elsewhere. */
int64_t binf_poly_leapfrog_workspace_bytes(int64_t C, int64_t K, int64_t N);
int32_t binf_poly_leapfrog_f64(double *q, double *p, const double *design,
                               const double *ys, double precision,
                               const double *precision_chain, void *workspace,
                               int64_t workspace_bytes, int64_t C, int64_t K,
                               int64_t N, double timestep, const double *dt_chain,
                               int32_t nsteps, int32_t mode, void *stream);

/* One HMCSampler.sample() (binf/samplers/hmc.py:136-164,183-191) for every
 * chain on the example's polynomial posterior with a SMALL or MEDIUM data set
 * (K <= 16 coefficients; N <= 1024 data points with a pairwise tree of height
 * <= 3 -- every N <= 920 and the multiples of 8 up to 1024; a chain's data are
 * spread over a group of 8 << H lanes, 8 lanes up to 128 points, one wave from
 * 513; mode | BINF_MODE_LANE_PER_CHAIN: one lane per chain, N <= 128; else
 * BINF_E_UNSUPPORTED), the whole transition in one launch (the per-step tier is
 * launch-bound there):
 *   log p(theta) = [lp_pre] + {prior, likelihood in the order prior_first says} + [lp_post]
 *   likelihood = -0.5 * sum((polyval(xs, theta) - ys)**2) * precision + N/2 * log(precision)
 *                (binf/example/likelihood.py:24-26,54-57; precision per chain if
 *                precision_chain != NULL)
 *   prior      = -0.5 * sum((theta - prior_means)**2 / prior_vars)
 *                (binf/example/priors.py:49-54; NULL, NULL = no prior)
 *   lp_pre / lp_post [C] or NULL: the posterior's theta-independent component
 *                terms that come before / after in its summation order
 *                (binf/pdf/posteriors.py:147-151).
 * The force is the likelihood's gradient only (the prior is registered
 * non-differentiable, posteriors.py:183).  Energies follow numpy's summation
 * order and are bit-identical to the per-step tier's; the force is an FMA dot
 * product (tolerance, like the MFMA path).  Other arguments as
 * binf_hmc_sample_gauss_f64. */
int32_t binf_hmc_sample_poly_f64(const double *q0, const double *p0,
                                 const double *u, double *q_out,
                                 uint8_t *accepted, int64_t *n_accepted,
                                 double *e_before, double *e_after,
                                 const double *xs, const double *ys,
                                 double precision, const double *precision_chain,
                                 const double *prior_means, const double *prior_vars,
                                 int32_t prior_first, const double *lp_pre,
                                 const double *lp_post, double timestep,
                                 double *dt_chain, int64_t C, int64_t K,
                                 int64_t N, int32_t nsteps, int32_t adapt,
                                 double uprate, double downrate, int32_t mode,
                                 void *stream);

/* ------------------------------------------------------------------------
 * The generic chain-rule contraction of the Likelihood plug-in surface,
 * Likelihood._evaluate_gradient (binf/pdf/likelihoods.py:148-155):
 *     return dfm.dot(emgrad)
 * for any forward model without a fused kernel of its own.
 *   batched == 0: jacobian [K x N], shared by all chains (linear forward models;
 *                 forwardmodels.py:23-28) -- two-operand f64 MFMA tiles;
 *   batched != 0: jacobian [C x K x N], one per chain -- streamed row products.
 *   emgrad [C x N] (error_model.gradient(mock_data=...), likelihoods.py:152-153),
 *   out [C x K]:  out[c,k] = sum_n jacobian[(c,) k, n] * emgrad[c, n].
 * The reference's BLAS summation order is not reproducible; this contraction is
 * held to 1e-10 * sum_n |J||r| (tests/poly_bounds.py).  Its own order depends on
 * (K, N) only: deterministic, and the same for any number of chains.
 * ---------------------------------------------------------------------- */
int32_t binf_jacobian_contract_f64(const double *jacobian, const double *emgrad,
                                   double *out, int64_t C, int64_t K, int64_t N,
                                   int32_t batched, void *stream);

/* out[i] = ((t_0[i] + t_1[i]) + t_2[i]) + ... : the Posterior's sums over its
 * components, Posterior._evaluate_log_prob (numpy.sum of a short list: sequential,
 * binf/pdf/posteriors.py:147-151) and _evaluate_gradient (posteriors.py:173-187),
 * in ONE launch and in the order given (the build's order: sorted component
 * name).  terms: HOST array of n_terms (<= 16) device vectors of n elements; a
 * NULL entry t stands for the host scalar scalars[t] (a component whose log-prob
 * is a Python float).  out may be one of the terms. */
int32_t binf_sum_terms_f64(const double *const *terms, const double *scalars,
                           int32_t n_terms, double *out, int64_t n, void *stream);

/* The same sum with a third kind of term: broadcast[t] != 0 (HOST array of n_terms
 * flags, or null = none) marks terms[t] as a pointer to ONE device double used for
 * every element -- a component whose log-prob is a 0-dim device tensor (e.g. the
 * GammaPrior of a precision shared by all chains) enters the sum without a device ->
 * host read-back.  A broadcast term must not lie inside out (BINF_E_ALIAS). */
int32_t binf_sum_terms_bcast_f64(const double *const *terms, const double *scalars,
                                 const uint8_t *broadcast, int32_t n_terms, double *out,
                                 int64_t n, void *stream);

/* ------------------------------------------------------------------------
 * n sweeps of the example's Gibbs loop in ONE launch:
 *     for i in range(n): gips.sample()          example_script.py:33-34
 * around GibbsSampler.sample (binf/samplers/gibbs.py:136-151; alphabetical
 * sweep: coefficients, then precision), for every chain, with a chain's state
 * in registers between the sweeps.  Per sweep and chain:
 *   coefficients  move == BINF_MOVE_HMC:  HMCSampler.sample (hmc.py:136-164,
 *                   183-191) on the conditional posterior of the coefficients
 *                   -- exactly the transition of binf_hmc_sample_poly_f64 with
 *                   precision_chain = the chain's current precision;
 *                 move == BINF_MOVE_RWMC: RWMCSampler.sample
 *                   (binf/example/samplers.py:78-92): proposal = state + change,
 *                   accept iff u < np.exp(lp_new - lp_old) (numpy's exp);
 *   precision     GammaSampler.sample (samplers.py:27-51):
 *                   rate = 0.5 * chi2(coefficients) + gamma_rate
 *                   precision = g / rate,  g ~ Gamma(gamma_shape, 1).
 * The conditional posterior of the coefficients is
 *   [gp] + {prior, likelihood in the order prior_first says} + [gp]
 *   gp = (gp_shape - 1) * log(precision) - precision * gp_rate: the GammaPrior
 *   term (binf/example/priors.py:23-25; a constant of the move), added first
 *   (gp_where == 1), last (2) or absent (0).
 * Draws: p0 / u / g supplied ([n x C x K], [n x C], [n x C]; parity with a host
 * stream), or NULL = generated in place from the Philox streams of binf_rng_*:
 *   p0[i][c][k] = element (chain_offset + c) * K + k of
 *                 binf_rng_normal_zig_f64 (zig != 0) / binf_rng_normal_f64 under
 *                 (seed_m, off_m + i * stride_m)          (HMC), or
 *                 -stepsize + 2 stepsize * (that element of binf_rng_uniform_f64)
 *                 as binf_rwmc_propose_f64 forms it         (RWMC);
 *   u[i][c]     = element chain_offset + c of binf_rng_uniform_f64 (seed_u, off_u + i * stride_u);
 *   g[i][c]     = element chain_offset + c of binf_rng_gamma_f64 (gamma_shape; seed_g, off_g + i * stride_g)
 *                 (generated only for gamma_shape >= 1, else BINF_E_UNSUPPORTED).
 * With samplers/rng.py:DeviceRNG serving all three: off_u = off_m + 1, off_g =
 * off_m + 2, strides 130 -- n sweeps then draw what n single sweeps draw.
 * Results are BIT-IDENTICAL to n single sweeps through binf_hmc_sample_poly_f64
 * (or binf_rwmc_*), binf_poly_gauss_logp_f64 and binf_gamma_precision_update_f64
 * with the same draws.  Records: the state after sweeps thin, 2 thin, ...
 * K <= 16, N <= 1024 with a pairwise tree of height <= 3, as
 * binf_hmc_sample_poly_f64.  struct_size = sizeof(binf_gibbs_poly_args)
 * (layout check; mismatch -> BINF_E_ARG).
 * ---------------------------------------------------------------------- */
#define BINF_MOVE_HMC  0
#define BINF_MOVE_RWMC 1
typedef struct binf_gibbs_poly_args {
    uint64_t struct_size;
    /* state, device */
    const double *coefficients;   /* [C x K] start                                  */
    const double *precision;      /* [C]     start                                  */
    double *coefficients_out;     /* [C x K] after sweep n; may be `coefficients`   */
    double *precision_out;        /* [C]     after sweep n; may be `precision`      */
    double *rec_coefficients;     /* [n / thin x C x K] or NULL                     */
    double *rec_precision;        /* [n / thin x C] or NULL                         */
    uint8_t *accepted;            /* [n x C] or NULL: the move's accept flags       */
    int64_t *n_accepted;          /* [C] or NULL, += accepted moves                 */
    double *e_before;             /* [n x C] or NULL (HMC)                          */
    double *e_after;              /* [n x C] or NULL (HMC)                          */
    /* model, device */
    const double *xs;             /* [N] */
    const double *ys;             /* [N] */
    const double *prior_means;    /* [K] or NULL (then prior_vars NULL too)         */
    const double *prior_vars;     /* [K] */
    /* supplied draws, device, or NULL */
    const double *p0;
    const double *u;
    const double *g;
    double *dt_chain;             /* [C] or NULL: per-chain HMC step sizes (in/out)  */
    double timestep;              /* HMC step size when dt_chain == NULL            */
    double uprate, downrate;      /* hmc.py:188-191                                  */
    double stepsize;              /* RWMC half-width                                 */
    double gp_shape, gp_rate;     /* GammaPrior term of the coefficient conditional  */
    double gamma_shape;           /* 0.5 N + prior.shape - 1, samplers.py:27-32       */
    double gamma_rate;            /* prior.rate of the precision conditional         */
    int64_t C, K, N;
    int64_t chain_offset;
    uint64_t seed_m, off_m, stride_m;
    uint64_t seed_u, off_u, stride_u;
    uint64_t seed_g, off_g, stride_g;
    int32_t move;                 /* BINF_MOVE_*                                     */
    int32_t mode;                 /* BINF_MODE_EXACT / BINF_MODE_FMA (HMC)           */
    int32_t nsteps;               /* leapfrog steps (HMC)                            */
    int32_t n;                    /* sweeps                                          */
    int32_t thin;
    int32_t n_adapt;              /* the first n_adapt HMC transitions adapt dt_chain */
    int32_t prior_first;
    int32_t gp_where;             /* 0 / 1 / 2, see above                            */
    int32_t zig;                  /* generated momenta: ziggurat (1) or Box-Muller (0) */
    int32_t keep_precision;       /* != 0: no precision draw -- n moves of the coefficients
                                     alone under a FIXED precision: n HMCSampler.sample()
                                     calls on the conditional posterior (g and the gamma
                                     arguments are ignored)                              */
} binf_gibbs_poly_args;
int32_t binf_gibbs_poly_sample_n_f64(const binf_gibbs_poly_args *args, void *stream);

/* GammaPrior._evaluate_log_prob (binf/example/priors.py:10-25), one value per chain:
 *   out[c] = (shape - 1) * log(precision[c]) - precision[c] * rate
 * (each operation rounded on its own, as the numpy / torch expression).
 * precision / out device [C] (out may alias precision). */
int32_t binf_gamma_logp_f64(const double *precision, double shape, double rate,
                            double *out, int64_t C, void *stream);

/* Conjugate precision draw, GammaSampler.sample (binf/example/samplers.py:34-51):
 * out[c] = g[c] / (-lp_unit[c] + prior_rate), g = supplied Gamma(shape) variates,
 * lp_unit = likelihood log-prob evaluated at precision = 1. */
int32_t binf_gamma_precision_update_f64(const double *g, const double *lp_unit,
                                        double prior_rate, double *out, int64_t C,
                                        void *stream);

/* ------------------------------------------------------------------------
 * Random-walk Metropolis subsampler, RWMCSampler.sample
 * (binf/example/samplers.py:78-92), as two launches around the pdf's evaluation
 * of the proposal.  Draws: supplied by the caller (parity with the reference's
 * np.random stream) or, when the pointer is NULL, generated from the Philox
 * stream of binf_rng_uniform_f64 under (seed, offset), keyed by GLOBAL index
 * (chain_offset = global index of chain 0 of this call).
 *
 * binf_rwmc_propose_f64 (samplers.py:81-83):
 *   proposal[c,k] = state[c,k] + change[c,k]
 *   change == NULL: change[c,k] = -stepsize + (stepsize - -stepsize) * U, U =
 *   element (chain_offset + c) * K + k of the uniform stream (seed, offset) --
 *   np.random.uniform(low=-stepsize, high=stepsize) as legacy numpy forms it.
 *   proposal may be exactly state (in place).
 *
 * binf_rwmc_accept_f64 (samplers.py:84-90):
 *   acc[c] = u[c] < np.exp(-(E_new[c] - E_old[c])),  E = -log_prob: lp_old / lp_new
 *   are the LOG-PROBABILITIES the pdf returned ([C]); numpy's exp, not csb's
 *   clipped one (overflow -> inf accepts, NaN rejects).
 *   u == NULL: u[c] = element chain_offset + c of the uniform stream (seed, offset).
 *   state_out[c,:] = acc ? proposal[c,:] : state[c,:]   (exactly proposal, exactly
 *   state, or disjoint);  accepted [C] or NULL;  n_accepted [C] or NULL, += acc
 *   (samplers.py:88).
 * ---------------------------------------------------------------------- */
int32_t binf_rwmc_propose_f64(const double *state, const double *change,
                              double *proposal, double stepsize, int64_t C,
                              int64_t K, uint64_t seed, uint64_t offset,
                              int64_t chain_offset, void *stream);
int32_t binf_rwmc_accept_f64(const double *proposal, const double *state,
                             const double *lp_old, const double *lp_new,
                             const double *u, double *state_out,
                             uint8_t *accepted, int64_t *n_accepted, int64_t C,
                             int64_t K, uint64_t seed, uint64_t offset,
                             int64_t chain_offset, void *stream);

/* ------------------------------------------------------------------------
 * Posterior-predictive density of a Gaussian error model over a grid of points,
 * from S drawn samples: predict (binf/example/misc.py:3-16) for every point of the
 * [nx x ny] grid that plot_prediction_tube walks with two Python loops
 * (binf/example/plots.py:8-11) -- one launch.
 *   out[i,j] = exp(log_sum_exp_s f[s,i,j]) / S                          (misc.py:16)
 *   f[s,i,j] = -0.5 * (mock[s,i] - ys[i,j])**2 * precision[s]
 *              + 0.5 * log(precision[s]) - half_log_2pi                  (misc.py:8)
 *   log_sum_exp(x) = log(sum(exp(x - max(x)))) + max(x)   (csb.numeric.log_sum_exp:
 *   csb is absent, its published definition; "parity unpinned")
 * mock [S x nx]: the forward model at predict_space[i] for sample s (for the
 * polynomial model: binf_poly_forward_f64 of the sampled coefficients); precision
 * [S]; ys, out [nx x ny]; half_log_2pi = 0.5 * log(2 pi) as the host computes it.
 * NaN / inf follow numpy (a negative precision gives NaN, S >= 1 required: the
 * reference's max() of no samples raises).  Lane-strided sums and the device
 * library's log / exp: ~1e-14 relative to the numpy restatement, not bit-identical.
 * workspace: binf_predictive_density_workspace_bytes(S, nx, ny) bytes of device
 * memory (0 for grids that fill the chip by themselves: NULL is then accepted) --
 * a small grid with many samples is computed chunk of samples by chunk and joined.
 * ---------------------------------------------------------------------- */
int64_t binf_predictive_density_workspace_bytes(int64_t S, int64_t nx, int64_t ny);
int32_t binf_predictive_density_f64(const double *mock, const double *precision,
                                    const double *ys, double *out, int64_t S,
                                    int64_t nx, int64_t ny, double half_log_2pi,
                                    void *workspace, int64_t workspace_bytes,
                                    void *stream);

/* ------------------------------------------------------------------------
 * Pairwise-distance-restraint model (BASELINE config C5; build-defined, the
 * reference has no code for it -- it follows the reference's forward-model /
 * error-model plug-in shape, binf/model/forwardmodels.py:10-66).
 * Coordinates x[c, 3*bead + axis]; pairs (i<j) in numpy.triu_indices order.
 * ---------------------------------------------------------------------- */

/* out[c,p] = sqrt(((x_i - x_j)**2).sum()) for p = (pair_i[p], pair_j[p]);
 * bit-identical to the numpy expression.  pair_i / pair_j device int32 [n_pairs]. */
int32_t binf_pairdist_forward_f64(const double *x, const int32_t *pair_i,
                                  const int32_t *pair_j, double *out, int64_t C,
                                  int64_t n_beads, int64_t n_pairs, void *stream);

/* Likelihood._evaluate_log_prob (binf/pdf/likelihoods.py:141-146) for the
 * pair-distance forward model + Gaussian error model, fused: the same bits as
 * binf_pairdist_forward_f64 followed by binf_gauss_err_logp_f64, without the
 * [C x n_pairs] distances going through HBM. */
/* workspace (ABI 6; the three entry points below): with FEWER chains than CUs -- or more than
 * 2048 beads -- a workgroup per chain walking the whole pair list leaves the chip idle (0.3 ms
 * at 1024 beads, 1.2 ms at 2048, 18 ms at 4096, whatever the number of chains).  np.sum adds
 * the sums of 8192-element chunks one after the other, so given
 * binf_pairdist_chi2_workspace_bytes(C, n_beads, n_pairs) bytes of device memory (0 when that
 * does not apply; NULL is always accepted -- only the speed changes) every chunk is a
 * workgroup of its own and a second launch adds the chunk sums in order: the same bits. */
int64_t binf_pairdist_chi2_workspace_bytes(int64_t C, int64_t n_beads, int64_t n_pairs);
int32_t binf_pairdist_gauss_logp_f64(const double *x, const int32_t *pair_i,
                                     const int32_t *pair_j, const double *ys,
                                     double precision, const double *precision_chain,
                                     double *out, int64_t C, int64_t n_beads,
                                     int64_t n_pairs, void *workspace,
                                     int64_t workspace_bytes, void *stream);

/* The same with a two-entry per-chain memo of chi^2 (it depends on the chain's coordinates
 * alone): memo_x [2 x C x 3 n_beads] / memo_chi2 [2 x C] / memo_state [2 x C] as in
 * binf_poly_gauss_logp_memo_f64 -- HMCSampler.sample() asks for the log-prob of the state
 * it ended the last transition with again as E_before (binf/samplers/hmc.py:148); that
 * state is the last proposal or the state before it, and both are in the memo. */
int32_t binf_pairdist_gauss_logp_memo_f64(const double *x, const int32_t *pair_i,
                                          const int32_t *pair_j, const double *ys,
                                          double precision, const double *precision_chain,
                                          double *out, double *memo_x, double *memo_chi2,
                                          uint8_t *memo_state, int64_t C, int64_t n_beads,
                                          int64_t n_pairs, void *workspace,
                                          int64_t workspace_bytes, void *stream);

/* HMCSampler.sample()'s energy (binf/samplers/hmc.py:143,148,150) for the restraint
 * posterior in ONE launch:
 *   energy[c]   = 0.5 * np.sum(p[c]**2) - log_prob[c]
 *   log_prob[c] = the Posterior's components added in their order, ((t0 + t1) + t2) + ...
 *                 (binf/pdf/posteriors.py:147-151).  term_kind[0 .. n_terms) (host array,
 *                 1 <= n_terms <= 4, each kind at most once) names them:
 *                   1  the restraint likelihood, -0.5 chi^2 precision_c + n_pairs/2 log
 *                      precision_c (as binf_pairdist_gauss_logp_f64) -- must be present;
 *                   0  an isotropic Gaussian prior, (-0.5 prior_k) * np.sum((x[c] - prior_x0)**2);
 *                   2, 3  a component whose variables are all fixed -- a constant of the
 *                      move, evaluated by the caller: extra0 / extra1 [C], or null and the
 *                      scalar (e.g. the GammaPrior of the precision inside a Gibbs sweep).
 * Replaces the per-step tier's row sum for the prior, memo check, chi^2 reduction,
 * binf_sum_terms_f64 and binf_hmc_energy_f64 (six launches) and is bit-identical to
 * them: every sum in numpy's order, every scalar operation in theirs.  x, p: device
 * [C x 3 n_beads]; energy [C]; log_prob [C] or null.  memo_x / memo_chi2 / memo_state: the
 * two-entry chi^2 memo of binf_pairdist_gauss_logp_memo_f64 (same buffers, same contents:
 * the two functions may share one memo), or all three null.  n_beads <= 2048. */
int32_t binf_pairdist_hmc_energy_f64(const double *x, const double *p, const int32_t *pair_i,
                                     const int32_t *pair_j, const double *ys, double precision,
                                     const double *precision_chain, double prior_k,
                                     double prior_x0, int32_t n_terms, const int32_t *term_kind,
                                     const double *extra0, double extra0_scalar,
                                     const double *extra1, double extra1_scalar,
                                     double *energy, double *log_prob, double *memo_x,
                                     double *memo_chi2, uint8_t *memo_state, int64_t C,
                                     int64_t n_beads, int64_t n_pairs, void *workspace, int64_t workspace_bytes,
                                     void *stream);

/* Energy gradient of the Gaussian restraint likelihood,
 *   out[c, 3i+a] = precision_c * sum_{j != i} (d_ij - ymat[j][i]) (x_i - x_j)[a] / d_ij,
 * i.e. Likelihood._evaluate_gradient (binf/pdf/likelihoods.py:148-155) without
 * forming the [3n x n(n-1)/2] Jacobian (coordinates in LDS; 32..256 beads: every
 * unordered pair once with its target distance in a register, otherwise
 * one-sided all-pairs loops).  Held to 1e-10 of the numpy expression; the
 * summation order depends on n_beads only, so a chain's result does not depend on
 * C.  ymat: device, SYMMETRIC [n_beads x n_beads] target distances (ymat[i][j] ==
 * ymat[j][i] is relied upon; the diagonal is ignored). */
int32_t binf_pairdist_gauss_grad_f64(const double *x, const double *ymat,
                                     double precision, const double *precision_chain,
                                     double *out, int64_t C, int64_t n_beads,
                                     void *stream);

/* The targets of the 32..256-bead force kernels in the order their waves hold them
 * (every unordered pair once, 32 targets per lane in registers): a function of
 * (ymat, n_beads) alone.  Packed once per model, the `_packed_` entry points below
 * read 32 coalesced values per lane at the start of a launch instead of walking the
 * [n x n] matrix tile by tile through LDS (once per workgroup: ~10 of the 175 us of
 * a 20-step trajectory at 256 chains).  binf_pairdist_packed_targets_bytes: size of
 * the packed form, 0 for bead counts the one-sided kernels serve (then there is
 * nothing to pack and `packed` must be null).  For 32..256 beads the results are the
 * same bits with and without `packed`; `packed` must have been made from the same ymat.
 * 257..1024 beads (ABI 6): the packed form is the step order of the ring kernels, which
 * compute every unordered pair once (2-4x the one-sided loops) and read their targets
 * from it for every force evaluation; WITH `packed` these bead counts take the ring
 * kernels, without it the one-sided loops -- the two agree to ~1e-15 relative (another
 * summation order), each is bit-identical between its force kernel and its fused
 * leapfrog and for any number of chains. */
int64_t binf_pairdist_packed_targets_bytes(int64_t n_beads);
int32_t binf_pairdist_pack_targets_f64(const double *ymat, double *packed, int64_t n_beads,
                                       void *stream);
/* workspace (ABI 6): for FEW chains of 257..1024 beads a workgroup per chain would leave most
 * of the chip idle; given binf_pairdist_tiles_workspace_bytes(C, n_beads) bytes of device
 * memory (0 when that does not apply: NULL is then fine, and it is always accepted -- only the
 * speed changes) every 64 x 64 tile of pairs becomes a wave of its own and a second launch adds
 * a bead's partial sums in the ring kernels' order: the same bits, ~5x at 32 chains of 1024
 * beads. */
int64_t binf_pairdist_tiles_workspace_bytes(int64_t C, int64_t n_beads);
int32_t binf_pairdist_gauss_grad_packed_f64(const double *x, const double *ymat,
                                            const double *packed, double precision,
                                            const double *precision_chain, double *out,
                                            int64_t C, int64_t n_beads, void *workspace,
                                            int64_t workspace_bytes, void *stream);

/* The whole leapfrog integration HMCSampler._leapfrog (binf/samplers/hmc.py:92-125)
 * for the restraint posterior in one launch: q, p device [C * 3n], integrated in
 * place over nsteps steps (half kick, (nsteps-1) x [drift, kick], drift, half
 * kick).  Gradient = precision_c * restraint force (+ optional isotropic
 * Gaussian prior term prior_k * (x - prior_x0), added before or after the
 * likelihood term as prior_first says -- the Posterior's component order).
 * Bit-identical to the per-step generic tier.  n_beads <= 1024. */
int32_t binf_pairdist_leapfrog_f64(double *q, double *p, const double *ymat,
                                   double precision, const double *precision_chain,
                                   int32_t has_prior, double prior_k, double prior_x0,
                                   int32_t prior_first, double timestep,
                                   const double *dt_chain, int32_t nsteps, int64_t C,
                                   int64_t n_beads, int32_t mode, void *stream);
/* the same with the packed targets (null: as binf_pairdist_leapfrog_f64) and, optionally,
 * the start positions read from q_from [C * 3n] instead of q (q then only receives the end
 * positions: the copy of the state HMCSampler.sample() makes before integrating,
 * hmc.py:140-141, is not needed); q_from null or == q: in place. */
int32_t binf_pairdist_leapfrog_packed_f64(double *q, const double *q_from, double *p,
                                          const double *ymat,
                                          const double *packed, double precision,
                                          const double *precision_chain, int32_t has_prior,
                                          double prior_k, double prior_x0, int32_t prior_first,
                                          double timestep, const double *dt_chain,
                                          int32_t nsteps, int64_t C, int64_t n_beads,
                                          int32_t mode, void *workspace,
                                          int64_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------
 * Device random draws (throughput mode): counter-based Philox4x32-10, key =
 * seed, element i of a call uses counter (i, offset).  Replace the
 * np.random.normal / uniform / gamma draws of binf/samplers/hmc.py:146,151 and
 * binf/example/samplers.py:47 when bit-parity with numpy's stream is not
 * required.  A caller advances `offset` by 1 per uniform / normal call and by
 * 128 per gamma call so that calls never reuse a counter.
 * `i` is a GLOBAL element index: out[l] receives element elem_offset + l
 * (elem_offset >= 0), so a rank filling its window of a larger logical array
 * (chains [s, s + C) of a [C_total x D] draw: elem_offset = s * D, n = C * D) writes
 * exactly what the unsharded call writes there.
 * ---------------------------------------------------------------------- */
int32_t binf_rng_uniform_f64(double *out, int64_t n, uint64_t seed, uint64_t offset,
                             int64_t elem_offset, void *stream);  /* [0, 1), 53 bits */
int32_t binf_rng_normal_f64(double *out, int64_t n, uint64_t seed, uint64_t offset,
                            int64_t elem_offset, void *stream);   /* Box-Muller       */
int32_t binf_rng_normal_zig_f64(double *out, int64_t n, uint64_t seed, uint64_t offset,
                                int64_t elem_offset, void *stream);  /* 1024-layer
                                                   ziggurat; offset < 2^48          */
/* The two draws of one HMC transition (hmc.py:146,151) in one launch: what
 * binf_rng_normal_zig_f64(normals, n_normals, seed, offset_normals, elem_offset_normals) and
 * binf_rng_uniform_f64(uniforms, n_uniforms, seed, offset_uniforms, elem_offset_uniforms)
 * write, bit for bit (the streams are those of the separate calls; their offsets must
 * differ). */
int32_t binf_rng_normal_zig_uniform_f64(double *normals, int64_t n_normals, double *uniforms,
                                        int64_t n_uniforms, uint64_t seed,
                                        uint64_t offset_normals, uint64_t offset_uniforms,
                                        int64_t elem_offset_normals,
                                        int64_t elem_offset_uniforms, void *stream);
int32_t binf_rng_gamma_f64(double *out, int64_t n, double shape, uint64_t seed,
                           uint64_t offset, int64_t elem_offset,
                           void *stream);                         /* Marsaglia-Tsang  */
/* Host-side evaluation of the generator's block function (known-answer tests). */
int32_t binf_rng_philox4x32_10(const uint32_t counter[4], const uint32_t key[2],
                               uint32_t out[4]);

/* ------------------------------------------------------------------------
 * Host-side helpers exposing the reduction geometry the kernels use, so that
 * CPU tests can check it against numpy's pairwise summation (no GPU needed).
 * ---------------------------------------------------------------------- */

/* Height of numpy's pairwise-sum recursion tree for a length-n vector
 * (0 when n <= 128). */
int32_t binf_pairwise_tree_height(int64_t n);

/* Leaf reached by the root-to-leaf path `path` (bit H-1 = first split, 0 =
 * left) in a tree padded to height H.  Outputs the leaf's offset and length,
 * its depth, and whether `path` is the canonical (lowest) path reaching it.
 * Returns 0, or BINF_E_ARG. */
int32_t binf_pairwise_leaf(int64_t n, int32_t H, int32_t path, int64_t *off,
                           int64_t *len, int32_t *depth, int32_t *canonical);

#ifdef __cplusplus
}
#endif
#endif /* BINF_HIP_H */

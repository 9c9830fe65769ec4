"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  numpy restatement of the
example application's Gibbs loop with HMC on the coefficients -- one chain,
exactly the operations the reference classes would perform:

  GibbsSampler.sample            binf/samplers/gibbs.py:136-151  (alphabetical
                                 sweep: 'coefficients', then 'precision')
  HMCSampler.sample              binf/samplers/hmc.py:136-164
  conditional Posterior          binf/pdf/posteriors.py:147-151,173-187,201-210
  GammaSampler.sample            binf/example/samplers.py:27-51
  priors of make_priors()        binf/example/priors.py:66-73

Parity status: the plumbing rules are pinned by the reference's test
known-answers (tests/test_host_mirror.py); ``example_script_chain`` (RWMC + Gamma,
the reference's own wiring) is pinned bit for bit by states the reference's own
RWMCSampler / GammaSampler classes produced (tests/golden/ref_example_chain_*.npz,
oracle/gen_ref_example.py); ``gibbs_hmc_chain`` inherits the status of
HMCSampler.sample: integrator pinned by reference output, energies + accept test
**parity unpinned** (csb's exp is absent).

Quirk Q6 is reproduced: the GammaPrior inside every CONDITIONAL posterior is a
clone built with (shape, shape) (binf/example/priors.py:27-32), so both the
HMC energy's constant term and the Gamma draw's rate use rate == shape == 1.0,
not the 0.2 that make_priors() constructs.
"""
import numpy as np

from oracle import ref_numpy as R

PRIOR_SHAPE = 1.0
PRIOR_RATE_CONSTRUCTED = 0.2          # binf/example/priors.py:69
PRIOR_RATE_IN_CONDITIONALS = 1.0      # == shape, quirk Q6


def example_data(n_data_points=20, seed=0):
    """example_script.py:17-26 after np.random.seed(seed)."""
    np.random.seed(seed)
    real_coeffs = np.array([2.0, -4.0, 1.0, 1.5])
    real_precision = 2.5
    xses = np.linspace(-2, 2, n_data_points)
    ys = np.random.normal(loc=R.polyval(xses, real_coeffs),
                          scale=1.0 / np.sqrt(real_precision))
    return xses, ys


def conditional_pdf(xses, ys, precision, K):
    return R.PolyCoefficientsConditional(
        xses, ys, precision, prior_means=np.zeros(K),
        prior_variances=np.ones(K) * 5, gamma_shape=PRIOR_SHAPE,
        gamma_rate=PRIOR_RATE_IN_CONDITIONALS)


def gibbs_hmc_chain(xses, ys, coeffs0, precision0, timestep, nsteps, p0, u, g):
    """Run len(p0) Gibbs sweeps for ONE chain with injected draws:
    p0[s] (momentum, [K]), u[s] (uniform), g[s] (Gamma(shape) variate).
    Returns per-sweep coefficients, precisions, accept flags, energies."""
    K = len(coeffs0)
    n = len(ys)
    coeffs = np.array(coeffs0, dtype=np.float64)
    tau = float(precision0)
    draws = {'i': 0}
    sampler = R.RefHMCSampler(None, coeffs.copy(), timestep, nsteps,
                              variable_name='coefficients',
                              normal=lambda size: p0[draws['i']].copy(),
                              uniform=lambda: u[draws['i']])
    out_c, out_t, out_a, out_eb, out_ea = [], [], [], [], []
    for s in range(len(p0)):
        draws['i'] = s
        # 'coefficients': HMC on the conditional posterior (precision fixed)
        sampler.pdf = conditional_pdf(xses, ys, tau, K)
        sampler.state = coeffs
        coeffs = sampler.sample()
        # 'precision': conjugate Gamma draw given the new coefficients
        shape = R.gamma_shape(n, PRIOR_SHAPE)
        rate = R.gamma_rate(xses, ys, coeffs, PRIOR_RATE_IN_CONDITIONALS)
        tau = R.gamma_draw(g[s], rate)
        out_c.append(coeffs.copy())
        out_t.append(tau)
        out_a.append(bool(sampler.last_move_accepted))
        out_eb.append(sampler.last_E_before)
        out_ea.append(sampler.last_E_after)
    return dict(coefficients=np.array(out_c), precision=np.array(out_t),
                accepted=np.array(out_a), e_before=np.array(out_eb),
                e_after=np.array(out_ea), gamma_shape=shape)


def example_script_chain(seed, sweeps, stepsize=0.1, n_data_points=20):
    """The reference's example_script.py as it stands (one chain, its own
    wiring: random-walk Metropolis on the coefficients + conjugate Gamma for the
    precision inside Gibbs), consuming ONE global legacy stream from
    ``np.random.seed(seed)`` on -- the data first (example_script.py:17-23), then
    per sweep, alphabetically (gibbs.py:141):

      'coefficients'  RWMCSampler.sample   binf/example/samplers.py:78-92
                      uniform(-step, step, size=K), then random()
      'precision'     GammaSampler.sample  binf/example/samplers.py:43-51
                      gamma(shape)

    Returns the data and the state after every sweep."""
    np.random.seed(seed)
    real_coeffs = np.array([2.0, -4.0, 1.0, 1.5])
    xses = np.linspace(-2, 2, n_data_points)
    ys = np.random.normal(loc=R.polyval(xses, real_coeffs), scale=1.0 / np.sqrt(2.5))
    coeffs = np.ones(4)
    tau = 1.0
    K = len(coeffs)
    n_moves = n_acc = 0
    out_c, out_t, out_a = [], [], []
    for _ in range(sweeps):
        pdf = conditional_pdf(xses, ys, tau, K)
        E_old = -pdf.log_prob(coefficients=coeffs)                       # :80
        change = np.random.uniform(low=-stepsize, high=stepsize, size=K)  # :81-82
        proposal = coeffs + change                                       # :83
        E_new = -pdf.log_prob(coefficients=proposal)                     # :84
        accepted = np.random.random() < np.exp(-(E_new - E_old))         # :86
        if accepted:
            coeffs = proposal
            n_acc += 1
        n_moves += 1
        shape = R.gamma_shape(n_data_points, PRIOR_SHAPE)
        rate = R.gamma_rate(xses, ys, coeffs, PRIOR_RATE_IN_CONDITIONALS)
        tau = np.random.gamma(shape) / rate                              # :47-49
        out_c.append(coeffs.copy())
        out_t.append(tau)
        out_a.append(bool(accepted))
    return dict(xs=xses, ys=ys, coefficients=np.array(out_c), precision=np.array(out_t),
                accepted=np.array(out_a), acceptance_rate=n_acc / float(n_moves))


# ---- the consumer side: posterior-predictive density (binf/example/misc.py, plots.py) ----------
def log_sum_exp(x, axis=0):
    """``csb.numeric.log_sum_exp`` -- csb is absent from /root/reference; this is its published
    definition (csb 1.2.x, csb/numeric/__init__.py): the maximum along ``axis`` taken out of the
    exponentials.  **Parity unpinned** (no reference output can be produced without csb)."""
    xmax = x.max(axis)
    return np.log(np.exp(x - xmax).sum(axis)) + xmax


def predict_integrands(x, y, coefficients, precisions, polynomial=R.polyval):
    """The list ``integrands`` of ``predict`` (binf/example/misc.py:8-9), one entry per sample;
    pinned bit for bit by tests/golden/ref_predict_*.npz (the reference's own statement run on
    data-only samples, oracle/gen_ref_predict.py)."""
    return np.array([-0.5 * (polynomial(x, c) - y) ** 2 * tau + 0.5 * np.log(tau)
                     - 0.5 * np.log(2.0 * np.pi) for c, tau in zip(coefficients, precisions)])


def predict(x, y, coefficients, precisions, polynomial=R.polyval):
    """``predict`` (binf/example/misc.py:3-16) for samples given as arrays
    (``coefficients`` [S x K], ``precisions`` [S])."""
    integrands = predict_integrands(x, y, coefficients, precisions, polynomial)
    return np.exp(log_sum_exp(integrands)) / len(coefficients)


def prediction_tube(coefficients, precisions, predict_space, ys_from, ys_to, n_ys, polynomial=R.polyval,
                    probs=None):
    """The numbers ``plot_prediction_tube`` draws (binf/example/plots.py:8-27): the y grid per x,
    the predictive density on it, its cumulative sums, the 5 % / 95 % limits and the trapezoid mean.
    ``probs``: densities to post-process instead of computing them (the fixtures pin the
    post-processing with densities handed to the reference's own statements)."""
    predicted_ys = np.array([np.linspace(ys_from[i], ys_to[i], n_ys) for i, _ in enumerate(predict_space)])
    if probs is None:
        probs = np.array([[predict(x, y, coefficients, precisions, polynomial) for y in predicted_ys[i]]
                          for i, x in enumerate(predict_space)])
    cdfs = np.cumsum(probs * (predicted_ys[:, 1] - predicted_ys[:, 0])[:, None], 1)
    lower = np.array([predicted_ys[i][np.where(cdfs[i] < 0.05)[0][-1]] for i in range(len(predict_space))])
    upper = np.array([predicted_ys[i][np.where(cdfs[i] > 0.95)[0][0]] for i in range(len(predict_space))])
    trapezoid = getattr(np, 'trapezoid', None) or np.trapz
    mean = np.array([trapezoid(predicted_ys[i] * probs[i], predicted_ys[i]) for i in range(len(predict_space))])
    return dict(predicted_ys=predicted_ys, probs=probs, cdfs=cdfs, lower=lower, upper=upper, prediction=mean)

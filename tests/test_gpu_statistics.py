"""Statistical correctness on the GPU: the samplers must sample their target
distribution.  Independent of the bitwise oracle (whose HMC numerics are not
pinned by the reference): a wrong accept rule, energy or
integrator would pass a restatement-vs-restatement comparison but not these.

All targets have closed forms:
* isotropic Gaussian N(x0, 1/k) -- C2's PDF, fused persistent kernel with the
  device RNG (every draw generated on the device);
* polynomial coefficients given the precision: Gaussian with precision matrix
  tau * A A^T + diag(1 / prior_var) -- the fused small-data transition;
* Gibbs over (coefficients, precision) -- checked through the conjugate
  identity E[tau | theta] averaged over the chain.
Tolerances are 6 standard errors of the respective Monte-Carlo mean (fixed
seeds, so the tests are deterministic)."""
import numpy as np
import pytest
import torch

from binf_amd.example.likelihood import POLYVAL, ForwardModel, GaussianErrorModel
from binf_amd.example.priors import GammaPrior, GaussianPrior
from binf_amd.pdf import IsotropicGaussian
from binf_amd.pdf.likelihoods import Likelihood
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('k,x0,mode', [(1.0, 0.0, 'exact'), (2.5, 0.3, 'exact'), (1.0, 0.0, 'fma')])
def test_gaussian_target_moments(device, k, x0, mode):
    C, D, L, dt = 2048, 64, 20, 0.3 / np.sqrt(k)
    rng = DeviceRNG(11, device)
    q0 = x0 + rng.normal((C, D), device) / np.sqrt(k)          # start in equilibrium
    s = HMCSampler(IsotropicGaussian(k, x0), q0, dt, L, variable_name='x', rng=rng, mode=mode)
    draws = s.sample_n(200, thin=10)                            # [20, C, D]
    assert 0.7 < float(s.acceptance_rate.mean()) < 1.0
    x = draws.reshape(-1, D).cpu().numpy()
    n = x.shape[0]                                              # chains are independent
    se_mean = 1.0 / np.sqrt(k * C)                              # per kept sweep: >= C independent values
    assert np.abs(x.mean(0) - x0).max() < 6 * se_mean
    var = ((x - x0) ** 2).mean(0)
    assert np.abs(var * k - 1.0).max() < 6 * np.sqrt(2.0 / C)
    # mixing: a long trajectory decorrelates the state from the start
    first, last = draws[0].cpu().numpy(), draws[-1].cpu().numpy()
    corr = np.mean((first - x0) * (last - x0)) * k
    assert abs(corr) < 6 / np.sqrt(C * D)
    assert n == 20 * C


def _conditional(xs, ys, K, tau, device):
    lik = Likelihood('points', ForwardModel(xs, POLYVAL), GaussianErrorModel(ys))
    post = Posterior({lik.name: lik},
                     {'precision_prior': GammaPrior(1.0, 0.2),
                      'coefficients_prior': GaussianPrior(np.zeros(K), np.full(K, 5.0))})
    return post, post.conditional_factory(precision=tau)


@pytest.mark.parametrize('fused', [True, False])
def test_polynomial_conditional_matches_its_analytic_gaussian(device, fused):
    """p(theta | tau, data) is Gaussian; the HMC force omits the prior (quirk
    Q4) but the energy has it, so the chain must still target the full
    conditional."""
    rs = np.random.RandomState(5)
    K, N, tau = 4, 20, 2.5
    xs = np.linspace(-2, 2, N)
    ys = np.polynomial.polynomial.polyval(xs, [2.0, -4.0, 1.0, 1.5]) + rs.standard_normal(N) / np.sqrt(tau)
    A = np.vstack([xs ** i for i in range(K)])
    P = tau * A @ A.T + np.eye(K) / 5.0
    cov = np.linalg.inv(P)
    mean = cov @ (tau * A @ ys)
    C = 4096 if fused else 256
    sweeps, burn = (120, 40) if fused else (40, 15)
    _, cond = _conditional(xs, ys, K, tau, device)
    rng = DeviceRNG(3, device)
    start = torch.from_numpy(mean + rs.standard_normal((C, K)) @ np.linalg.cholesky(cov).T).to(device)
    s = HMCSampler(cond, start, 0.02, 50, variable_name='coefficients', rng=rng)
    s.fused_transition = fused
    assert (s._fused_spec('coefficients', K) is not None) == fused
    kept = []
    for i in range(sweeps):
        x = s.sample()
        if i >= burn and i % 5 == 0:
            kept.append(x.clone())
    acc = float(s.acceptance_rate.mean())
    assert 0.6 < acc <= 1.0
    x = torch.stack(kept).cpu().numpy()                         # [n, C, K]
    se = np.sqrt(np.diag(cov) / C)                              # one sweep of C chains at least
    assert (np.abs(x.mean((0, 1)) - mean) < 6 * se).all()
    emp = np.cov(x.reshape(-1, K).T)
    assert np.abs(emp - cov).max() < 0.15 * np.abs(cov).max()


def _load_example(name):
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        'examples', name + '.py')
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_example_gaussian_chains_forgets_its_start(device):
    """examples/gaussian_chains.py: over-dispersed start (sd 3), the second
    half of the run must have the target's moments."""
    out = _load_example('gaussian_chains').main(
        ['--chains', '512', '--dims', '96', '--draws', '640', '--thin', '32',
         '--k', '2.0', '--x0', '0.5', '--timestep', '0.25'])
    assert out.shape == (20, 512, 96)
    tail = out[10:]
    assert abs(float(tail.mean()) - 0.5) < 0.01
    assert abs(float(tail.var()) - 0.5) < 0.01


def test_example_distance_restraints_keeps_the_structure(device):
    """examples/distance_restraints.py (C5's model at a small size): the
    sampled structures reproduce the target distances to about the noise."""
    kept = _load_example('distance_restraints').main(
        ['--chains', '24', '--beads', '48', '--iterations', '60', '--thin', '10'])
    assert kept.shape == (6, 24, 144)
    assert torch.isfinite(kept).all()
    kept = _load_example('distance_restraints').main(
        ['--chains', '8', '--beads', '40', '--iterations', '40', '--thin', '10', '--gibbs'])
    assert kept.shape == (4, 8, 120) and torch.isfinite(kept).all()
    # beyond 256 beads with the few chains structure inference runs with: the symmetric force as a
    # wave per tile, chi^2 by chunks (700 beads: 11 blocks, ghosts in the last one)
    kept = _load_example('distance_restraints').main(
        ['--chains', '6', '--beads', '700', '--iterations', '40', '--thin', '10', '--timestep', '0.001'])
    assert kept.shape == (4, 6, 2100) and torch.isfinite(kept).all()
    x = kept[-1].reshape(6, 700, 3)
    assert float((x[:, 1:] - x[:, :-1]).norm(dim=-1).mean()) > 0.5        # a structure, not a collapsed point


def test_example_custom_pdf_samples_both_wells(device):
    """examples/custom_pdf.py: a duck-typed torch pdf (double well) on the
    per-step tier; coordinates sit near +-1, half of them in each well."""
    s = _load_example('custom_pdf').main(
        ['--chains', '512', '--dims', '32', '--draws', '300'])
    x = s.state
    assert 0.5 < float(s.acceptance_rate.mean()) <= 1.0
    assert abs(float((x > 0).double().mean()) - 0.5) < 0.02
    assert 0.8 < float(x.abs().mean()) < 1.1
    # the same run with every transition replayed from one HIP graph: the same chains
    g = _load_example('custom_pdf').main(['--chains', '512', '--dims', '32', '--draws', '300', '--graph'])
    assert len(g._graphs) == 1 and torch.equal(g.state, x) and torch.equal(g.n_accepted, s.n_accepted)


def test_gibbs_launches_with_two_different_moves_agree_on_the_posterior(device):
    """The multi-sweep Gibbs launch (csrc/gibbs_poly.hip) with its two coefficient
    moves -- HMC and the reference's random walk -- shares only the energy code:
    trajectories, proposals, accept rules (clipped exp vs np.exp) and draw streams
    differ.  Both must sample the SAME joint posterior of the example; and given the
    sampled coefficients, the precision draws must follow their conjugate Gamma
    (mean = shape / rate), which does not involve either move."""
    from binf_amd.example.misc import make_posterior
    from binf_amd.example.samplers import make_hmc_sampler, make_sampler
    from binf_amd.samplers import BinfState
    np.random.seed(0)
    xs = np.linspace(-2, 2, 20)
    ys = np.random.normal(loc=POLYVAL(xs, np.array([2.0, -4.0, 1.0, 1.5])), scale=1.0 / np.sqrt(2.5))
    C = 1024

    def start():
        return BinfState(dict(coefficients=torch.ones((C, 4), dtype=torch.float64, device=device),
                              precision=torch.ones(C, dtype=torch.float64, device=device)))
    hmc = make_hmc_sampler(make_posterior(xs, ys, POLYVAL), 0.02, 50, start(), rng=DeviceRNG(1, device))
    rw = make_sampler(make_posterior(xs, ys, POLYVAL), 0.1, start(), rng=DeviceRNG(2, device))
    hmc.sample_n(300, record=False)                    # burn-in
    rw.sample_n(30000, record=False)
    a = hmc.sample_n(400, thin=20)                     # 20 kept sweeps x 1024 chains
    b = rw.sample_n(40000, thin=2000)
    for key in ('coefficients', 'precision'):
        xa = a[key].reshape(-1, a[key].shape[-1] if key == 'coefficients' else 1).cpu().numpy()
        xb = b[key].reshape(-1, xa.shape[1]).cpu().numpy()
        # chains are independent; kept sweeps of a chain are nearly so: use C as the
        # effective sample size of each mean (conservative)
        se = np.sqrt(xa.var(axis=0) / C + xb.var(axis=0) / C)
        assert np.all(np.abs(xa.mean(axis=0) - xb.mean(axis=0)) < 6 * se), (key, xa.mean(0), xb.mean(0), se)
        assert np.all(np.abs(np.log(xa.std(axis=0) / xb.std(axis=0))) < 0.15), key
    # conjugate identity, per kept state of the HMC run: tau ~ Gamma(shape, rate(theta)) with
    # shape = 0.5 n + alpha - 1 = 10, rate = 0.5 chi^2(theta) + 1 (the conditional's rate, Q6)
    th = a['coefficients'].reshape(-1, 4).cpu().numpy()
    tau = a['precision'].reshape(-1).cpu().numpy()
    rate = 0.5 * np.sum((POLYVAL(xs, th.T) - ys[None, :]) ** 2, axis=1) + 1.0
    z = tau * rate                                     # ~ Gamma(10, 1): mean 10, variance 10
    assert abs(z.mean() - 10.0) < 6 * np.sqrt(10.0 / len(z))
    assert abs(z.var() - 10.0) < 6 * 10.0 * np.sqrt(2.0 / len(z) + 0.6 / len(z))


@pytest.mark.parametrize('n', [24, 48])
def test_gibbs_over_coordinates_and_precision_recovers_the_noise_level(device, n):
    """The restraint model inside the reference's Gibbs scheme (HMC on the coordinates, the
    conjugate Gamma draw of the precision, one precision per chain): the sampled precision
    settles at the precision the target distances were generated with -- the plug-in
    surface, the per-chain precisions of the fused kernels and the chi^2 memo (shared by
    the HMC energies and the Gamma update) together.  24 beads: one-sided force loops;
    48: one wave per block pair."""
    from binf_amd.example.distance import make_distance_likelihood, make_restraint_gibbs_sampler
    from binf_amd.samplers import BinfState
    C, tau_true = 64, 25.0
    rs = np.random.RandomState(n)
    truth = np.cumsum(rs.standard_normal((n, 3)), axis=0) * 0.6
    I, J = np.triu_indices(n, 1)
    d = np.sqrt(((truth[I] - truth[J]) ** 2).sum(1))
    ys = np.abs(d + rs.standard_normal(d.shape) / np.sqrt(tau_true))
    lik = make_distance_likelihood(ys, n)
    post = Posterior({lik.name: lik},
                     {'coordinates_prior': IsotropicGaussian(0.01, 0.0, name='coordinates_prior',
                                                             variable_name='coordinates'),
                      'precision_prior': GammaPrior(1.0, 0.2)})
    rng = DeviceRNG(7, device)
    start = BinfState({'coordinates': torch.from_numpy(truth.reshape(1, -1)).to(device)
                       + 0.05 * rng.normal((C, 3 * n), device),
                       'precision': torch.full((C,), 1.0, dtype=torch.float64, device=device)})
    gips = make_restraint_gibbs_sampler(post, 0.01, 10, start, rng=rng, timestep_adaption_limit=150)
    taus, cond_means = [], []
    n_pairs = len(ys)
    # GammaSampler (binf/example/samplers.py:27-51): shape = N/2 + alpha - 1, rate = chi^2/2 + the
    # CONDITIONAL prior's rate, which is its shape (quirk Q6: GammaPrior.clone)
    shape, prior_rate = 0.5 * n_pairs + 1.0 - 1.0, 1.0
    for it in range(400):
        st = gips.sample()
        if it >= 200 and it % 5 == 0:
            taus.append(st.variables['precision'].clone())
            chi2 = -2.0 * lik.log_prob(coordinates=st.variables['coordinates'], precision=1.0)
            cond_means.append(shape / (0.5 * chi2 + prior_rate))
    taus = torch.stack(taus).cpu().numpy()
    cond_means = torch.stack(cond_means).cpu().numpy()
    acc = float(gips.subsamplers['coordinates'].acceptance_rate.mean())
    assert 0.3 < acc <= 1.0
    # conjugate identity: E[tau] = E[ E[tau | coordinates] ], Monte-Carlo error of the sample mean
    rel_sd = 1.0 / np.sqrt(shape)
    n_eff = taus.size
    assert abs(taus.mean() / cond_means.mean() - 1.0) < 6.0 * rel_sd / np.sqrt(n_eff)
    # ... and the level itself: the data's precision, pulled down by the prior rate against
    # chi^2 / 2 ~ N / (2 tau) and up by the 3n coordinates that absorb residuals
    assert 0.6 * tau_true < taus.mean() < 1.3 * tau_true, taus.mean()
    assert taus.std(axis=0).mean() < 4.0 * taus.mean() * rel_sd

"""The HIP kernels of the example model and the example's sampling loop against outputs of THE
REFERENCE'S OWN example code (``tests/golden/ref_example_*.npz``, ``oracle/gen_ref_example.py``; see
``tests/test_ref_example.py`` for what ran and how).

* Horner forward model, error-model gradient, chi^2 (log-prob at unit precision), Gaussian prior:
  **bit for bit**; log-probs that involve ``log(precision)``: 1e-13; the chain-rule gradient (BLAS in
  the reference): 1e-10 of its sum-of-magnitudes scale; the design matrix: bit for bit.
* ``example_script.py`` itself, one chain on the reference's host stream (RWMC + Gamma inside Gibbs):
  every state equal, bit for bit, to what the reference's own ``RWMCSampler`` / ``GammaSampler``
  produced."""
import numpy as np
import pytest
import torch

from binf_amd import _native
from binf_amd.example.likelihood import POLYVAL, ForwardModel, GaussianErrorModel, make_likelihood
from binf_amd.example.misc import make_posterior
from binf_amd.example.priors import GammaPrior, GaussianPrior
from binf_amd.example.samplers import make_sampler
from binf_amd.samplers import BinfState
from conftest import golden_files, load_golden

pytestmark = pytest.mark.gpu
ident = lambda p: p.split('ref_example_')[-1][:-4]


def dev_t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)


@pytest.mark.parametrize('path', golden_files('ref_example_models_'), ids=ident)
def test_model_kernels_vs_the_reference_run(device, path):
    g = load_golden(path)
    xs, ys, theta, taus = g['xs'], g['ys'], g['theta'], g['precision']
    C, K = theta.shape
    N = len(xs)
    st = int(g['mock_stride'])
    tth, tys, ttau = dev_t(theta, device), dev_t(ys, device), dev_t(taus, device)
    fwm, em = ForwardModel(xs, POLYVAL), GaussianErrorModel(ys)
    mock = fwm(coefficients=tth)                                        # binf_poly_forward_f64
    assert np.array_equal(mock.cpu().numpy()[:, ::st], g['mock'])
    J = fwm.jacobi_matrix(coefficients=tth).cpu().numpy()
    assert np.array_equal(J if K * N <= 40000 else J[:, ::97], g['jacobi'])
    eg = em.gradient(mock_data=mock, precision=ttau).cpu().numpy()
    assert np.array_equal(eg[:, ::st], g['error_grad'])
    lik = make_likelihood(xs, ys, POLYVAL)
    # unit precision: log(1) = 0 exactly, the rest is numpy's arithmetic in numpy's order
    lp1 = lik.log_prob(coefficients=tth, precision=1.0).cpu().numpy()
    assert np.array_equal(lp1, g['error_logp_unit_precision'])
    lp1_unfused = em.log_prob(mock_data=mock, precision=1.0).cpu().numpy()
    assert np.array_equal(lp1_unfused, g['error_logp_unit_precision'])
    lp = lik.log_prob(coefficients=tth, precision=ttau).cpu().numpy()
    assert np.allclose(lp, g['error_logp'], rtol=1e-13, atol=0)
    gr = lik.gradient(coefficients=tth, precision=ttau).cpu().numpy()
    Jn = np.vstack([xs ** i for i in range(K)])
    for c in range(C):
        bound = np.abs(Jn).dot(np.abs((POLYVAL(xs, theta[c]) - ys) * taus[c]))
        assert np.all(np.abs(gr[c] - g['likelihood_grad'][c]) <= 1e-10 * bound), c
    gp = GaussianPrior(g['prior_means'], g['prior_variances'])
    assert np.array_equal(gp.log_prob(coefficients=tth).cpu().numpy(), g['gaussian_prior_logp'])
    gam = GammaPrior(float(g['gamma_prior_shape']), float(g['gamma_prior_rate']))
    got = gam.log_prob(precision=ttau).cpu().numpy()
    assert np.allclose(got, g['gamma_prior_logp'], rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize('path', golden_files('ref_example_chain_'), ids=ident)
def test_example_script_chain_vs_the_reference_subsamplers(device, path):
    g = load_golden(path)
    seed, stepsize, n = int(g['seed']), float(g['stepsize']), len(g['precision'])
    N = len(g['xs'])
    np.random.seed(seed)
    xs = np.linspace(-2, 2, N)
    ys = np.random.normal(loc=POLYVAL(xs, np.array([2.0, -4.0, 1.0, 1.5])), scale=1.0 / np.sqrt(2.5))
    assert np.array_equal(ys, g['ys'])
    start = BinfState(dict(coefficients=dev_t(np.ones((1, 4)), device), precision=dev_t(np.ones(1), device)))
    gips = make_sampler(make_posterior(xs, ys, POLYVAL), stepsize, start)
    for s in range(n):
        st = gips.sample()
        assert np.array_equal(st.variables['coefficients'].cpu().numpy()[0], g['coefficients'][s]), s
        assert float(st.variables['precision'].cpu().numpy()[0]) == g['precision'][s], s
    rate = gips.last_draw_stats['coefficients'].acceptance_rate
    assert abs(float(rate) - g['acceptance_rate'][-1]) < 1e-12
    assert type(gips.last_draw_stats['coefficients'])._fields[0] == str(g['last_draw_stats_field'])

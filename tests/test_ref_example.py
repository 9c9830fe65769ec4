"""The oracle's example-model arithmetic against outputs of THE REFERENCE'S OWN example code.

``tests/golden/ref_example_*.npz`` were produced in the build container by
``oracle/gen_ref_example.py``: the method bodies of the reference's polynomial forward model,
Gaussian error model and the two priors (``binf/example/likelihood.py:24-30,54-61``,
``priors.py:23-25,49-54``) compiled unchanged from its syntax tree and called with a data-only
``self``; its ``RWMCSampler`` class (``samplers.py:54-92``) executed as it stands; its
``GammaSampler._calculate_shape / _calculate_rate / sample`` (``samplers.py:27-51``) as they stand
(the Python-2-only ``_get_prior`` replaced by a method that returns the prior).  csb is absent and
nothing is substituted for it.  They are glued by the reference's own Likelihood / Posterior log-prob
method bodies (``binf/pdf/likelihoods.py:122-155``, ``posteriors.py:117-151``) over duck-typed
components; the generating script supplies the component order (sorted name, Q5), the completion of
fixed variables and the Gibbs sweep.

CPU: the numpy restatement reproduces every file bit for bit.  The HIP kernels are held to the same
files in ``tests/test_gpu_ref_example.py``."""
import numpy as np
import pytest

from conftest import golden_files, load_golden
from oracle import ref_example as RE
from oracle import ref_numpy as R

MODELS = golden_files('ref_example_models_')
CHAINS = golden_files('ref_example_chain_')
ident = lambda p: p.split('ref_example_')[-1][:-4]


def test_fixture_set():
    assert len(MODELS) == 5 and len(CHAINS) == 2
    for f in MODELS + CHAINS:
        prov = str(load_golden(f)['provenance'])
        assert 'REFERENCE' in prov and 'samplers.py:54-92' in prov and 'csb absent' in prov


@pytest.mark.parametrize('path', MODELS, ids=ident)
def test_restated_model_arithmetic_reproduces_the_reference_bitwise(path):
    g = load_golden(path)
    xs, ys, theta, taus = g['xs'], g['ys'], g['theta'], g['precision']
    C, K = theta.shape
    st = int(g['mock_stride'])
    for c in range(C):
        pdf = R.PolyCoefficientsConditional(xs, ys, taus[c], g['prior_means'], g['prior_variances'],
                                            float(g['gamma_prior_shape']), float(g['gamma_prior_rate']))
        mock = R.polyval(xs, theta[c])                                 # likelihood.py:24-26
        assert np.array_equal(mock[::st], g['mock'][c])
        comps = pdf.component_log_probs(theta[c])
        assert comps['points'] == g['error_logp'][c]                   # likelihood.py:54-57
        assert comps['coefficients_prior'] == g['gaussian_prior_logp'][c]   # priors.py:49-54
        assert comps['precision_prior'] == g['gamma_prior_logp'][c]    # priors.py:23-25
        assert np.array_equal(((mock - ys) * taus[c])[::st], g['error_grad'][c])   # likelihood.py:59-61
        # the chain rule (likelihoods.py:155) goes through BLAS in both: same numpy, same bits here
        assert np.array_equal(pdf.gradient(coefficients=theta[c]), g['likelihood_grad'][c])
        # the Gamma draw's rate: -log_prob at unit precision + prior rate (samplers.py:34-41)
        assert R.gamma_rate(xs, ys, theta[c], 0.0) == -g['error_logp_unit_precision'][c]
    J = pdf.jacobi_matrix(theta[0])                                    # likelihood.py:28-30
    assert np.array_equal(J if K * len(xs) <= 40000 else J[:, ::97], g['jacobi'])


@pytest.mark.parametrize('path', CHAINS, ids=ident)
def test_restated_example_loop_reproduces_the_reference_subsamplers_bitwise(path):
    """oracle/ref_example.py:example_script_chain -- RWMC on the coefficients + conjugate Gamma draw of
    the precision, one global legacy stream from the seed on -- against the states the reference's own
    RWMCSampler.sample and GammaSampler.sample produced, sweep by sweep."""
    g = load_golden(path)
    n = len(g['precision'])
    out = RE.example_script_chain(int(g['seed']), n, stepsize=float(g['stepsize']), n_data_points=len(g['xs']))
    assert np.array_equal(out['xs'], g['xs']) and np.array_equal(out['ys'], g['ys'])
    assert np.array_equal(out['coefficients'], g['coefficients'])
    assert np.array_equal(out['precision'], g['precision'])
    assert np.array_equal(out['accepted'], g['accepted'])
    assert out['acceptance_rate'] == g['acceptance_rate'][-1]
    assert R.gamma_shape(len(g['xs']), RE.PRIOR_SHAPE) == float(g['gamma_shape'])   # the "- 1" of samplers.py:32
    assert 0.05 < g['accepted'].mean() < 0.95


@pytest.mark.parametrize('path', MODELS, ids=ident)
def test_c_restatement_of_the_polynomial_model_reproduces_the_reference_bitwise(path):
    """oracle/oracle_c.c: oracle_polyval / oracle_poly_gauss_logp -- the checker of the Python-free
    host program (tests/cabi/host_check.cpp) -- against what the reference's own ForwardModel /
    GaussianErrorModel code produced (likelihood.py:24-26, 54-57): mock data bit for bit, the
    log-prob at unit precision bit for bit (log(1) = 0), at the fixture's precisions to an ulp of
    the N/2 log(precision) term (C's log and numpy's are different functions)."""
    from oracle import c_oracle
    g = load_golden(path)
    xs, ys, theta, taus = g['xs'], g['ys'], g['theta'], g['precision']
    st = int(g['mock_stride'])
    assert np.array_equal(c_oracle.polyval(xs, theta)[:, ::st], g['mock'])
    lp1, chi2 = c_oracle.poly_gauss_logp(theta, xs, ys, 1.0)
    assert np.array_equal(lp1, g['error_logp_unit_precision'])
    assert np.array_equal(-0.5 * chi2 * 1.0 + 0.0, lp1)
    lp, _ = c_oracle.poly_gauss_logp(theta, xs, ys, taus)
    logz = np.abs(len(xs) * 0.5 * np.log(taus))
    assert np.all(np.abs(lp - g['error_logp']) <= 4e-16 * np.maximum(logz, np.abs(lp)))

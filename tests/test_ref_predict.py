"""The consumer side of the sample store -- ``predict`` (``binf/example/misc.py:3-16``) and the
numbers of ``plot_prediction_tube`` (``binf/example/plots.py:8-27``) -- restated in
``oracle/ref_example.py`` and held to outputs of THE REFERENCE'S OWN statements
(``tests/golden/ref_predict_*.npz``, ``oracle/gen_ref_predict.py``: the integrand block of
``predict`` and the tube's y grid / cdfs / limits / trapezoid mean, run unchanged out of the
reference's syntax tree).  ``csb.numeric.log_sum_exp`` is absent: its definition, and with it
the last line of ``predict``, stays "parity unpinned"; an independent check (the mean of the
Gaussian densities, which is what the expression equals) stands in for it.

The HIP kernel is held to the restatement in ``tests/test_gpu_predict.py``."""
import numpy as np
import pytest

from conftest import golden_files, load_golden
from oracle import ref_example as E

FILES = golden_files('ref_predict_')


def test_fixture_set_says_where_it_comes_from():
    assert len(FILES) == 3
    for f in FILES:
        prov = str(load_golden(f)['provenance'])
        assert 'REFERENCE' in prov and 'misc.py:7-14' in prov and 'plots.py:8-9,12-16,20-23' in prov
        assert 'NOT run' in prov and 'csb' in prov


@pytest.mark.parametrize('path', FILES, ids=lambda p: p.split('ref_predict_')[-1][:-4])
def test_restatement_reproduces_the_reference_integrands_bitwise(path):
    g = load_golden(path)
    for x, y, want in zip(g['pts_x'], g['pts_y'], g['integrands']):
        got = E.predict_integrands(x, y, g['coefficients'], g['precisions'])
        assert np.array_equal(got, want)


@pytest.mark.parametrize('path', FILES, ids=lambda p: p.split('ref_predict_')[-1][:-4])
def test_restatement_reproduces_the_reference_tube_numbers_bitwise(path):
    g = load_golden(path)
    t = E.prediction_tube(g['coefficients'], g['precisions'], g['predict_space'], g['ys_from'], g['ys_to'],
                          int(g['n_ys']), probs=g['probs_in'])
    for name in ('predicted_ys', 'cdfs', 'lower', 'upper', 'prediction'):
        assert np.array_equal(t[name], g[name]), name
    # the densities handed to the reference's statements were the restatement's own
    full = E.prediction_tube(g['coefficients'], g['precisions'], g['predict_space'], g['ys_from'],
                             g['ys_to'], int(g['n_ys']))
    assert np.array_equal(full['probs'], g['probs_in'])
    assert np.all(full['lower'] < full['prediction']) and np.all(full['prediction'] < full['upper'])


def test_predict_is_the_mean_of_the_gaussian_densities():
    """exp(log_sum_exp(f)) / S with f the log of N(y | m_s, 1/tau_s) IS the sample mean of those
    densities: an independent reading of misc.py:16 that needs no log_sum_exp at all."""
    rs = np.random.RandomState(5)
    S, K = 300, 4
    c = rs.standard_normal((S, K))
    tau = rs.gamma(3.0, 1.0, size=S)
    for x, y in rs.standard_normal((8, 2)):
        m = np.array([E.R.polyval(x, ci) for ci in c])
        want = np.mean(np.sqrt(tau / (2 * np.pi)) * np.exp(-0.5 * tau * (m - y) ** 2))
        assert abs(E.predict(x, y, c, tau) - want) <= 1e-13 * want


def test_log_sum_exp_edge_cases():
    assert E.log_sum_exp(np.array([-1000.0, -1000.0])) == -1000.0 + np.log(2.0)
    assert E.log_sum_exp(np.array([700.0, 710.0])) > 710.0           # no overflow
    with np.errstate(invalid='ignore'):
        assert np.isnan(E.log_sum_exp(np.array([-np.inf, -np.inf])))  # inf - inf, as numpy has it
    with pytest.raises(ValueError):
        E.log_sum_exp(np.array([]))

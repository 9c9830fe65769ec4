"""The one logarithm the log-probs take per chain -- N/2 log(precision) of the Gaussian error
model (binf/example/likelihood.py:55), (shape - 1) log(precision) of the Gamma prior
(binf/example/priors.py:23-25) -- is CORRECTLY ROUNDED on the device (csrc/log_cr.hpp: double-double
2 atanh series): a fixed bit pattern, the double nearest the true logarithm, checked here against
mpmath at 200 bits.  numpy's own log is within an ulp of it (equal for ~99.9 % of arguments on this
host; it differs between CPUs), the device library's log for ~98 %."""
import numpy as np
import pytest
import torch

from binf_amd import _native

pytestmark = pytest.mark.gpu


def _log_cr(t, device):
    # (shape - 1) * log(tau) - tau * rate with shape = 2, rate = 0: the logarithm itself
    return _native.gamma_logp(torch.from_numpy(np.ascontiguousarray(t)).to(device), 2.0, 0.0).cpu().numpy()


def test_log_of_the_precision_is_correctly_rounded(device):
    mp = pytest.importorskip('mpmath')
    mp.mp.prec = 200
    rs = np.random.RandomState(5)
    t = np.concatenate([rs.uniform(0.3, 6.0, 6000), np.exp(rs.uniform(np.log(1e-3), np.log(1e3), 6000)),
                        np.exp(rs.uniform(-700.0, 700.0, 3000)), 1.0 + rs.uniform(-1e-3, 1e-3, 2000),
                        1.0 + rs.uniform(-1e-12, 1e-12, 500), 2.0 ** rs.randint(-1000, 1000, 300).astype(np.float64),
                        [1.0, 2.0, 0.5, 2.5, 4.0, 1e-300, 1e300, 0.7071067811865476, 0.7071067811865475,
                         1.4142135623730951, 1.414213562373095, 2.2250738585072014e-308, 1.7976931348623157e308]])
    want = np.array([float(mp.log(mp.mpf(float(x)))) for x in t])      # mpf -> float rounds to nearest
    got = _log_cr(t, device)
    assert np.array_equal(got, want), int((got != want).sum())
    # numpy's log agrees almost always, and never by more than an ulp
    off = np.log(t) != want
    assert off.mean() < 0.01
    assert np.all(np.abs(np.log(t) - want) <= np.spacing(np.abs(want)))


def test_log_special_values_follow_numpy(device):
    with np.errstate(all='ignore'):
        t = np.array([0.0, -0.0, -1.0, np.nan, 5e-324, 1e-310])
        got = _log_cr(t, device)
        want = np.log(t)
    assert got[0] == -np.inf and got[1] == -np.inf and np.isnan(got[2]) and np.isnan(got[3])
    assert np.all(np.abs(got[4:] - want[4:]) <= 2 * np.spacing(np.abs(want[4:])))    # subnormals: the library's log


def test_the_log_prob_of_the_polynomial_likelihood_with_a_correctly_rounded_log(device):
    """-0.5 chi^2 tau + N/2 log(tau): chi^2 and the scaling are numpy's bits (np.sum order), the log
    term the correctly rounded one -- so the whole log-prob equals the numpy expression evaluated
    with a correctly rounded log, for ANY precision, per chain or one for the batch."""
    mp = pytest.importorskip('mpmath')
    mp.mp.prec = 200
    from numpy.polynomial.polynomial import polyval
    rs = np.random.RandomState(11)
    C, K, N = 64, 5, 700
    xs, ys = rs.uniform(-1, 1, N), rs.standard_normal(N)
    co = rs.standard_normal((C, K))
    tau = rs.uniform(0.2, 9.0, C)
    dev_t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    got = _native.poly_gauss_logp(dev_t(co), dev_t(xs), dev_t(ys), dev_t(tau)).cpu().numpy()
    want = np.array([-0.5 * np.sum((polyval(xs, co[c]) - ys) ** 2) * tau[c]
                     + N * 0.5 * float(mp.log(mp.mpf(float(tau[c])))) for c in range(C)])
    assert np.array_equal(got, want)
    one = _native.poly_gauss_logp(dev_t(co), dev_t(xs), dev_t(ys), 2.7).cpu().numpy()
    want1 = np.array([-0.5 * np.sum((polyval(xs, co[c]) - ys) ** 2) * 2.7 + N * 0.5 * float(mp.log(mp.mpf(2.7)))
                      for c in range(C)])
    assert np.array_equal(one, want1)

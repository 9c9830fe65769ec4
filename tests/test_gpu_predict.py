"""``binf_predictive_density_f64`` and its host mirror (``binf_amd/example/misc.py``:
``predict`` / ``predict_grid`` / ``prediction_tube``) against the numpy restatement of
``binf/example/misc.py:3-16`` and ``plots.py:8-27`` (``oracle/ref_example.py``, whose integrand
and tube post-processing are pinned by reference output -- tests/test_ref_predict.py).

Floating-point bar: 1e-12 relative on every density (the kernel sums the samples lane-strided,
not in numpy's pairwise order, and uses the device library's log / exp; measured ~1e-15), the
Horner values underneath are bit-identical; the 5 % / 95 % limits are grid values and must be
EQUAL."""
import ctypes

import numpy as np
import pytest
import torch

from binf_amd import _native
from binf_amd.example import misc
from binf_amd.example.likelihood import POLYVAL
from binf_amd.samplers import BinfState
from conftest import golden_files, load_golden
from oracle import ref_example as E

pytestmark = pytest.mark.gpu

RTOL = 1e-12


def dev_t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)


def close(got, want):
    got, want = np.asarray(got), np.asarray(want)
    return np.all(np.abs(got - want) <= RTOL * np.abs(want))


@pytest.mark.parametrize('path', golden_files('ref_predict_'), ids=lambda p: p.split('ref_predict_')[-1][:-4])
def test_tube_and_points_against_the_restatement(device, path):
    g = load_golden(path)
    samples = (dev_t(g['coefficients'], device), dev_t(g['precisions'], device))
    t = misc.prediction_tube(samples, POLYVAL, g['predict_space'], g['ys_from'], g['ys_to'], int(g['n_ys']))
    assert np.array_equal(t.predicted_ys, g['predicted_ys'])
    assert close(t.probs, g['probs_in']) and t.probs.min() >= 0.0
    assert np.array_equal(t.lower, g['lower']) and np.array_equal(t.upper, g['upper'])
    assert close(t.cdfs, g['cdfs']) and close(t.prediction, g['prediction'])
    for x, y in zip(g['pts_x'], g['pts_y']):
        want = E.predict(x, y, g['coefficients'], g['precisions'])
        got = misc.predict(x, y, samples, POLYVAL)
        assert isinstance(got, float) and abs(got - want) <= RTOL * want


@pytest.mark.parametrize('S,K,nx,ny', [(1, 1, 1, 1), (5, 3, 1, 7), (16, 4, 16, 8), (17, 4, 17, 9),
                                       (257, 33, 33, 150), (3000, 4, 100, 31),
                                       # few points, many samples: the samples are cut into chunks
                                       (5000, 4, 3, 7), (511, 4, 1, 1), (512, 4, 1, 1), (20000, 4, 33, 17)])
def test_grid_shapes_against_the_restatement(device, S, K, nx, ny):
    rs = np.random.RandomState(S + nx)
    c = rs.standard_normal((S, K)) * 0.3
    tau = rs.gamma(4.0, 0.5, size=S)
    xs = rs.uniform(-1, 1, size=nx)
    ys = rs.standard_normal((nx, ny)) * 2
    got = misc.predict_grid(xs, ys, (dev_t(c, device), dev_t(tau, device)), POLYVAL).cpu().numpy()
    stride = max(1, (nx * ny) // 40)                 # the restatement loops in Python: a sample of the grid
    for flat in range(0, nx * ny, stride):
        i, j = divmod(flat, ny)
        want = E.predict(xs[i], ys[i, j], c, tau)
        assert abs(got[i, j] - want) <= RTOL * want, (i, j)
    # every point: the mean of the Gaussian densities (vectorised numpy, no log_sum_exp)
    m = np.array([[E.R.polyval(x, ci) for x in xs] for ci in c])                       # [S, nx]
    dens = np.sqrt(tau / (2 * np.pi))[:, None, None] * np.exp(-0.5 * tau[:, None, None] *
                                                              (m[:, :, None] - ys[None]) ** 2)
    assert np.all(np.abs(got - dens.mean(0)) <= 1e-11 * dens.mean(0) + 1e-300)


def test_states_of_a_sampling_loop_and_chain_batched_states(device):
    """``samples`` as the reference's loop collects them (a list of BinfState), one chain or a
    batch of chains per state: every chain of every state is one sample."""
    rs = np.random.RandomState(9)
    c = rs.standard_normal((12, 4)) * 0.2
    tau = rs.gamma(4.0, 0.5, size=12)
    single = [BinfState(dict(coefficients=dev_t(c[i], device), precision=float(tau[i]))) for i in range(12)]
    batched = [BinfState(dict(coefficients=dev_t(c[i:i + 4], device), precision=dev_t(tau[i:i + 4], device)))
               for i in (0, 4, 8)]
    want = E.predict(0.3, -0.1, c, tau)
    a, b = misc.predict(0.3, -0.1, single, POLYVAL), misc.predict(0.3, -0.1, batched, POLYVAL)
    assert a == b and abs(a - want) <= RTOL * want
    with pytest.raises(ValueError):
        misc.predict(0.3, -0.1, [], POLYVAL)
    with pytest.raises(ValueError):
        misc.predict_grid([0.0, 1.0], [[0.0]], single, POLYVAL)


def test_a_users_polynomial_callable_is_applied_as_given(device):
    rs = np.random.RandomState(10)
    c, tau = rs.standard_normal((50, 3)) * 0.2, rs.gamma(4.0, 0.5, size=50)
    calls = []

    def cubic_free(xs, coefficients):                 # c0 + c1 x + c2 x^2 through the library's Horner
        calls.append(tuple(coefficients.shape))
        return _native.poly_forward(coefficients, xs)

    xs, ys = np.linspace(-1, 1, 5), rs.standard_normal((5, 6))
    samples = (dev_t(c, device), dev_t(tau, device))
    got = misc.predict_grid(xs, ys, samples, cubic_free)
    assert calls == [(50, 3)] and torch.equal(got, misc.predict_grid(xs, ys, samples, POLYVAL))


def test_nan_and_inf_follow_numpy(device):
    c = np.zeros((4, 2))
    xs, ys = np.array([0.0, 1.0]), np.array([[0.0, 0.5], [1.0, -1.0]])
    with np.errstate(all='ignore'):
        for tau in ([1.0, -1.0, 2.0, 1.0],            # log of a negative precision: NaN everywhere
                    [0.0, 0.0, 0.0, 0.0],              # every term -inf: inf - inf = NaN, as numpy has it
                    [0.0, 1.0, 0.0, 4.0]):             # -inf terms drop out of the sum
            tau = np.array(tau)
            got = misc.predict_grid(xs, ys, (dev_t(c, device), dev_t(tau, device)), POLYVAL).cpu().numpy()
            want = np.array([[E.predict(xs[i], ys[i, j], c, tau) for j in range(2)] for i in range(2)])
            assert np.array_equal(np.isnan(got), np.isnan(want))
            ok = ~np.isnan(want)
            assert np.all(np.abs(got[ok] - want[ok]) <= RTOL * np.abs(want[ok]))
    # many samples on a small grid (chunks of samples joined by a second launch): a chunk whose
    # terms are all -inf drops out, all chunks -inf give NaN, one NaN term poisons the point
    S = 4000
    assert _native.lib().binf_predictive_density_workspace_bytes(S, 2, 2) > 0
    cs = np.zeros((S, 2))
    with np.errstate(all='ignore'):
        for tau in (np.where(np.arange(S) < 1500, 0.0, 2.0), np.zeros(S),
                    np.where(np.arange(S) == 3999, -1.0, 1.0), np.where(np.arange(S) % 2 == 0, 0.0, 3.0)):
            got = misc.predict_grid(xs, ys, (dev_t(cs, device), dev_t(tau, device)), POLYVAL).cpu().numpy()
            want = np.array([[E.predict(xs[i], ys[i, j], cs, tau) for j in range(2)] for i in range(2)])
            assert np.array_equal(np.isnan(got), np.isnan(want))
            ok = ~np.isnan(want)
            assert np.all(np.abs(got[ok] - want[ok]) <= RTOL * np.abs(want[ok]))
    # far tails underflow to 0 without NaN
    got = misc.predict_grid([0.0], [[1e6]], (dev_t(c, device), dev_t(np.ones(4), device)), POLYVAL)
    assert float(got[0, 0]) == 0.0


def test_c_abi_refusals(device):
    lib = _native.lib()
    mock = dev_t(np.zeros((4, 3)), device)
    tau = dev_t(np.ones(4), device)
    ys = dev_t(np.zeros((3, 5)), device)
    out = torch.empty((3, 5), dtype=torch.float64, device=device)
    h = 0.5 * np.log(2 * np.pi)
    args = lambda m, t, y, o, S=4, nx=3, ny=5, ws=None, wb=0: (m, t, y, o, S, nx, ny, h, ws, wb, None)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    assert lib.binf_predictive_density_f64(*args(p(mock), p(tau), p(ys), p(out), S=0)) == _native.E_ARG
    assert lib.binf_predictive_density_f64(*args(None, p(tau), p(ys), p(out))) == _native.E_ARG
    assert lib.binf_predictive_density_f64(*args(p(mock), p(tau), p(ys), p(ys))) == _native.E_ALIAS
    assert lib.binf_predictive_density_f64(*args(p(mock), p(tau), p(ys), p(out), nx=0)) == 0
    assert lib.binf_predictive_density_f64(*args(p(mock), p(tau), p(ys), p(out))) == 0
    torch.cuda.synchronize()
    assert torch.all(out > 0)
    # many samples on a small grid need the workspace the library asks for
    S = 4096
    big, taus = dev_t(np.zeros((S, 3)), device), dev_t(np.ones(S), device)
    need = lib.binf_predictive_density_workspace_bytes(S, 3, 5)
    assert need > 0 and lib.binf_predictive_density_workspace_bytes(4, 3, 5) == 0
    ws = torch.empty(need // 8, dtype=torch.float64, device=device)
    assert lib.binf_predictive_density_f64(*args(p(big), p(taus), p(ys), p(out), S=S)) == _native.E_ARG
    assert lib.binf_predictive_density_f64(*args(p(big), p(taus), p(ys), p(out), S=S, ws=p(ws), wb=need - 8)) == _native.E_ARG
    assert lib.binf_predictive_density_f64(*args(p(big), p(taus), p(ys), p(out), S=S, ws=p(out), wb=need)) == _native.E_ALIAS
    assert lib.binf_predictive_density_f64(*args(p(big), p(taus), p(ys), p(out), S=S, ws=p(ws), wb=need)) == 0
    torch.cuda.synchronize()
    assert torch.all(out > 0)

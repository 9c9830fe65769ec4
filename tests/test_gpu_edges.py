"""GPU edge cases through the C ABI: empty batches, single chains, ragged chain
counts (not a multiple of chains-per-wave / wave / workgroup), many chains,
extreme values."""
import numpy as np
import pytest
import torch

from binf_amd import _native
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
from oracle import c_oracle

pytestmark = pytest.mark.gpu


def dev_t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)


def test_empty_batches_are_no_ops(device):
    e2 = torch.empty((0, 64), dtype=torch.float64, device=device)
    e1 = torch.empty(0, dtype=torch.float64, device=device)
    acc = torch.empty(0, dtype=torch.uint8, device=device)
    _native.hmc_sample_gauss(e2, e2.clone(), e1, e2.clone(), acc, None, None, None,
                             0.1, None, 3, 1.0, 0.0, False, 1.05, 0.95)
    assert _native.row_sum(e2).shape == (0,)
    _native.leapfrog_kick(e2, e2.clone(), 0.1)
    _native.leapfrog_drift(e2, e2.clone(), 0.1)
    assert _native.gauss_grad(e2, 1.0, 0.0).shape == (0, 64)
    s = HMCSampler(IsotropicGaussian(), e2, 0.1, 3, variable_name='x')
    assert s.sample(p0=e2.clone(), u=e1).shape == (0, 64)
    torch.cuda.synchronize()


@pytest.mark.parametrize('D,C', [(1024, 1), (1024, 3), (1024, 5), (512, 7), (33, 1), (33, 9),
                                 (33, 33), (4, 1), (4, 63), (4, 65), (8, 100001)])
def test_ragged_chain_counts_fused_and_persistent(device, D, C):
    rs = np.random.RandomState(C + D)
    L, dt, n = 3, 0.4, 3
    q0 = rs.standard_normal((C, D))
    p0 = rs.standard_normal((n, C, D))
    u = rs.uniform(size=(n, C))
    want = q0
    for i in range(n):
        r = c_oracle.hmc_sample_gauss(want, p0[i], u[i], dt, L, nthreads=8)
        want = r['q_out']
    a = HMCSampler(IsotropicGaussian(), dev_t(q0, device), dt, L, variable_name='x')
    for i in range(n):
        out = a.sample(p0=dev_t(p0[i], device), u=dev_t(u[i], device))
    b = HMCSampler(IsotropicGaussian(), dev_t(q0, device), dt, L, variable_name='x')
    rec = b.sample_n(n, p0=dev_t(p0, device), u=dev_t(u, device))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    assert np.array_equal(rec[-1].cpu().numpy(), want)
    assert np.array_equal(b.last_move_accepted.cpu().numpy(), r['accepted'].astype(bool))


def test_single_chain_one_dimensional_state_like_the_reference(device):
    """The reference's own shape: state [D], one chain."""
    D, L, dt = 1024, 20, 0.2
    rs = np.random.RandomState(1)
    q0, p0, u = rs.standard_normal(D), rs.standard_normal(D), rs.uniform(size=1)
    s = HMCSampler(IsotropicGaussian(), dev_t(q0, device), dt, L, variable_name='x')
    out = s.sample(p0=dev_t(p0, device), u=dev_t(u, device))
    want = c_oracle.hmc_sample_gauss(q0[None], p0[None], u, dt, L)
    assert out.shape == (D,)
    assert np.array_equal(out.cpu().numpy(), want['q_out'][0])


def test_extreme_values(device):
    C, D, L = 16, 1024, 4
    rs = np.random.RandomState(2)
    q0 = rs.standard_normal((C, D))
    p0 = rs.standard_normal((C, D))
    u = rs.uniform(size=C)
    q0[0] *= 1e150            # energies overflow to inf
    q0[1] *= 1e-300           # subnormal squares
    q0[2, :] = 0.0
    p0[2, :] = 0.0            # exactly stationary: dE = 0, accept iff u < 1
    q0[3, 7] = np.inf
    u[4] = 0.0                # always accepted
    with np.errstate(all='ignore'):
        want = c_oracle.hmc_sample_gauss(q0, p0, u, 0.1, L)
    q_out = torch.empty((C, D), dtype=torch.float64, device=device)
    acc = torch.empty(C, dtype=torch.uint8, device=device)
    eb = torch.empty(C, dtype=torch.float64, device=device)
    ea = torch.empty(C, dtype=torch.float64, device=device)
    _native.hmc_sample_gauss(dev_t(q0, device), dev_t(p0, device), dev_t(u, device), q_out,
                             acc, None, eb, ea, 0.1, None, L, 1.0, 0.0, False, 1.05, 0.95)
    assert np.array_equal(acc.cpu().numpy(), want['accepted'])
    assert np.array_equal(q_out.cpu().numpy(), want['q_out'], equal_nan=True)
    assert np.array_equal(eb.cpu().numpy(), want['e_before'], equal_nan=True)
    assert np.array_equal(ea.cpu().numpy(), want['e_after'], equal_nan=True)
    assert want['accepted'][2] == 1 and want['accepted'][4] == 1


def test_row_sum_many_rows_and_long_rows(device):
    rs = np.random.RandomState(3)
    x = rs.standard_normal((70001, 5))
    got = _native.row_sum(dev_t(x, device)).cpu().numpy()
    assert np.array_equal(got, np.array([np.sum(r) for r in x]))
    y = rs.standard_normal((2, 300007))
    got = _native.row_sum(dev_t(y, device), _native.ROW_SUMSQ).cpu().numpy()
    assert np.array_equal(got, np.array([np.sum(r ** 2) for r in y]))


@pytest.mark.parametrize('C,D', [(1, 1), (7, 5), (33, 129), (4, 1024), (3, 8200), (2, 20000)])
def test_hmc_energy_is_the_three_step_expression_bitwise(device, C, D):
    """binf_hmc_energy_f64 against hmc.py:143,148: -log_prob + 0.5 * np.sum(p**2),
    including signed zeros and infinities of the log-prob."""
    rs = np.random.RandomState(C + D)
    p = rs.standard_normal((C, D))
    lp = rs.standard_normal(C) * 100.0
    lp[0] = -np.inf if C > 2 else lp[0]
    if C > 3:
        lp[1], lp[2] = 0.0, -0.0
        p[2] = 0.0
    got = _native.hmc_energy(dev_t(p, device), dev_t(lp, device)).cpu().numpy()
    want = np.array([-lp[c] + 0.5 * np.sum(p[c] ** 2) for c in range(C)])
    assert np.array_equal(got, want)
    assert np.array_equal(np.signbit(got), np.signbit(want))
    with pytest.raises(ValueError):
        _native.hmc_energy(dev_t(p, device), dev_t(lp[:-1] if C > 1 else np.zeros(2), device))


def test_non_contiguous_inputs_are_accepted(device):
    """A transposed state / momentum view and a user PDF that returns a
    non-contiguous gradient must give the same result as contiguous data."""
    C, D, L, dt = 12, 64, 3, 0.3
    rs = np.random.RandomState(9)
    q0, p0, u = rs.standard_normal((C, D)), rs.standard_normal((C, D)), rs.uniform(size=C)
    want = c_oracle.hmc_sample_gauss(q0, p0, u, dt, L)
    q_t = dev_t(q0.T.copy(), device).t()            # [C, D] view with swapped strides
    p_t = dev_t(p0.T.copy(), device).t()
    assert not q_t.is_contiguous()
    s = HMCSampler(IsotropicGaussian(), q_t, dt, L, variable_name='x')
    out = s.sample(p0=p_t, u=dev_t(u, device))
    assert np.array_equal(out.cpu().numpy(), want['q_out'])

    class Strided(object):
        def log_prob(self, x):
            return -0.5 * _native.row_sum(x, _native.ROW_SUMSQ)

        def gradient(self, x):
            return x.t().contiguous().t()            # same values, non-contiguous

    s2 = HMCSampler(Strided(), dev_t(q0, device), dt, L, variable_name='x')
    out2 = s2.sample(p0=dev_t(p0, device), u=dev_t(u, device))
    assert np.array_equal(out2.cpu().numpy(), want['q_out'])


def test_clipped_exp_matches_numpy_to_an_ulp(device):
    """binf_clipped_exp_f64 = csb.numeric.exp = exp(clip(x, -308, 709)); the
    same device function decides every accept test in the library."""
    rs = np.random.RandomState(0)
    x = np.concatenate([rs.uniform(-320, 720, 200000), rs.uniform(-1, 1, 100000),
                        rs.standard_normal(100000) * 1e-8,
                        [-1e300, -308.0, -307.9999, 0.0, -0.0, 1.0, 708.9, 709.0, 710.0, 1e300,
                         np.inf, -np.inf]])
    got = _native.clipped_exp(dev_t(x, device)).cpu().numpy()
    want = np.exp(np.clip(x, -308.0, 709.0))
    assert np.isfinite(got).all() and (got > 0).all()
    ulp = np.abs(got - want) / np.spacing(want)
    assert ulp.max() <= 1.0, ulp.max()
    assert (ulp == 0).mean() > 0.9
    assert got[x == 0.0].tolist() == [1.0, 1.0]
    assert np.isnan(_native.clipped_exp(dev_t([np.nan], device)).cpu().numpy()[0])

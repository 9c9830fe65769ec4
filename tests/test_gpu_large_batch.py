"""Batches beyond 2**32 elements (34 GB per ``[C x D]`` tensor): the sizes 288 GB of HBM invite
and the place where a 32-bit index in a kernel, a launcher or a generator would show.

No oracle finishes at this size; the checks are size-independent properties (the prompt's "at
full sizes ... through properties the domain offers"): chains never interact, and every draw is
keyed by the GLOBAL chain index, so any WINDOW of the big batch -- the first chains, the chains
whose elements straddle flat index 2**32, the last chains -- recomputed on its own as a small
batch (bit-identical to the oracle at such sizes: tests/test_gpu_hmc_gauss.py) must reproduce
the big run's rows bit for bit."""
import numpy as np
import pytest
import torch

from binf_amd import _native
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG

pytestmark = pytest.mark.gpu

D = 1024
EDGE = (1 << 32) // D                    # the chain whose first element has flat index 2**32
C_BIG = EDGE + 4096                      # 4 198 400 chains: 4.3e9 elements, 34.4 GB
GIB = float(1 << 30)


def windows(C, w=300):
    return [(0, w), (EDGE - w, EDGE + w), (C - w, C)]


def need(device, gib):
    free, _ = torch.cuda.mem_get_info(device)
    if free < gib * GIB:
        pytest.skip('needs %.0f GiB of free HBM, %.0f free' % (gib, free / GIB))


@pytest.fixture(autouse=True)
def release():
    yield
    torch.cuda.empty_cache()


def test_fused_gaussian_supplied_draws_beyond_2_32_elements(device):
    need(device, 120)
    g = torch.Generator(device=device).manual_seed(1)
    q0 = torch.randn((C_BIG, D), dtype=torch.float64, device=device, generator=g)
    p0 = torch.randn((C_BIG, D), dtype=torch.float64, device=device, generator=g)
    u = torch.rand(C_BIG, dtype=torch.float64, device=device, generator=g)
    s = HMCSampler(IsotropicGaussian(), q0, 0.21, 2, variable_name='x', record_energies=True)
    out = s.sample(p0=p0, u=u)
    acc, eb, ea = s.last_move_accepted, s.last_e_before, s.last_e_after
    assert out.shape == (C_BIG, D) and 0.5 < float(acc.double().mean()) < 1.0
    assert bool(torch.isfinite(ea).all())
    for a, b in windows(C_BIG):
        w = HMCSampler(IsotropicGaussian(), q0[a:b].clone(), 0.21, 2, variable_name='x',
                       record_energies=True)
        wout = w.sample(p0=p0[a:b].clone(), u=u[a:b].clone())
        assert torch.equal(wout, out[a:b]), (a, b)
        assert torch.equal(w.last_move_accepted, acc[a:b])
        assert torch.equal(w.last_e_before, eb[a:b]) and torch.equal(w.last_e_after, ea[a:b])
    # a rejected chain keeps its row, an accepted one moved -- over the whole batch
    moved = (out != q0).any(dim=1)
    assert torch.equal(moved, acc.bool())


def test_fused_gaussian_in_kernel_draws_beyond_2_32_elements(device):
    """``sample_n`` with the draws made inside the kernel: the stream of a lane is a function of
    its GLOBAL chain index (64-bit), so a window run with ``chain_offset`` draws the same."""
    need(device, 85)
    g = torch.Generator(device=device).manual_seed(2)
    q0 = torch.randn((C_BIG, D), dtype=torch.float64, device=device, generator=g)
    s = HMCSampler(IsotropicGaussian(2.5, 0.3), q0, 0.12, 2, variable_name='x', rng=DeviceRNG(77, device))
    s.sample_n(2, record=False)
    out, nacc = s.state, s.n_accepted
    assert 0.5 < float(nacc.double().mean()) / 2 <= 1.0
    for a, b in windows(C_BIG):
        w = HMCSampler(IsotropicGaussian(2.5, 0.3), q0[a:b].clone(), 0.12, 2, variable_name='x',
                       rng=DeviceRNG(77, device, chain_offset=a))
        w.sample_n(2, record=False)
        assert torch.equal(w.state, out[a:b]), (a, b)
        assert torch.equal(w.n_accepted, nacc[a:b])


def test_per_step_kernels_beyond_2_32_elements(device):
    """The generic tier's kernels (gradient, kick + drift, np.sum-order row reductions, the
    stand-alone generator) on a 34 GB batch against the same kernels on windows."""
    need(device, 120)
    rng = DeviceRNG(5, device)
    q = rng.normal((C_BIG, D), device)
    p = rng.normal((C_BIG, D), device)
    big = {'q0': [q[a:b].clone() for a, b in windows(C_BIG)], 'p0': [p[a:b].clone() for a, b in windows(C_BIG)]}
    # the generator itself: element i of the draw is a function of its global flat index
    for (a, b), want in zip(windows(C_BIG), big['q0']):
        shard = DeviceRNG(5, device, chain_offset=a)
        assert torch.equal(shard.normal((b - a, D), device), want), (a, b)
    dt = torch.rand(C_BIG, dtype=torch.float64, device=device) * 0.1 + 0.05
    grad = _native.gauss_grad(q, 2.5, 0.3)
    _native.leapfrog_kick_drift(q, p, grad, 99.0, dt_chain=dt)
    e = _native.row_sum(q, _native.ROW_SUMSQ_SHIFT, shift=0.3, scale=-1.25)
    k = _native.row_sum(p, _native.ROW_SUMSQ)
    for (a, b), q0, p0 in zip(windows(C_BIG), big['q0'], big['p0']):
        g = _native.gauss_grad(q0, 2.5, 0.3)
        assert torch.equal(g, grad[a:b])
        _native.leapfrog_kick_drift(q0, p0, g, 99.0, dt_chain=dt[a:b].clone())
        assert torch.equal(q0, q[a:b]) and torch.equal(p0, p[a:b]), (a, b)
        assert torch.equal(_native.row_sum(q0, _native.ROW_SUMSQ_SHIFT, shift=0.3, scale=-1.25), e[a:b])
        assert torch.equal(_native.row_sum(p0, _native.ROW_SUMSQ), k[a:b])


def test_polynomial_forward_and_log_prob_beyond_2_32_elements(device):
    """C3's data length with enough chains that the mock data [C x N] crosses 2**32 elements
    (the fused log-prob never materialises it; the forward model does)."""
    need(device, 60)
    N, K = 16384, 33
    edge = (1 << 32) // N
    C = edge + 1024                                   # 263 168 chains x 16384 data: 34.5 GB of mock data
    rs = np.random.RandomState(3)
    xs = torch.from_numpy(np.linspace(-1, 1, N)).to(device)
    ys = torch.from_numpy(rs.standard_normal(N)).to(device)
    theta = torch.from_numpy(rs.standard_normal((C, K)) * 0.3).to(device)
    tau = torch.from_numpy(rs.gamma(2.0, 1.0, size=C)).to(device)
    mock = _native.poly_forward(theta, xs)
    lp = _native.poly_gauss_logp(theta, xs, ys, tau)
    chi = _native.row_sumsq_diff(mock, ys)
    assert bool(torch.isfinite(lp).all())
    for a, b in [(0, 64), (edge - 64, edge + 64), (C - 64, C)]:
        t = theta[a:b].clone()
        m = _native.poly_forward(t, xs)
        assert torch.equal(m, mock[a:b]), (a, b)
        assert torch.equal(_native.poly_gauss_logp(t, xs, ys, tau[a:b].clone()), lp[a:b])
        assert torch.equal(_native.row_sumsq_diff(m, ys), chi[a:b])


def test_long_chains_beyond_2_32_elements(device):
    """D = 16384 (the long-chain kernels, csrc/hmc_gauss_big.hip: 8192-element chunks, several
    launches per transition) with enough chains to cross 2**32 elements."""
    need(device, 125)
    Dl = 16384
    edge = (1 << 32) // Dl
    C = edge + 64                                     # 262 208 chains x 16384
    g = torch.Generator(device=device).manual_seed(4)
    q0 = torch.randn((C, Dl), dtype=torch.float64, device=device, generator=g)
    p0 = torch.randn((C, Dl), dtype=torch.float64, device=device, generator=g)
    u = torch.rand(C, dtype=torch.float64, device=device, generator=g)
    s = HMCSampler(IsotropicGaussian(), q0, 0.05, 2, variable_name='x', record_energies=True)
    out = s.sample(p0=p0, u=u)
    acc, ea = s.last_move_accepted, s.last_e_after
    assert 0.3 < float(acc.double().mean()) < 1.0
    for a, b in [(0, 8), (edge - 8, edge + 8), (C - 8, C)]:
        w = HMCSampler(IsotropicGaussian(), q0[a:b].clone(), 0.05, 2, variable_name='x', record_energies=True)
        assert torch.equal(w.sample(p0=p0[a:b].clone(), u=u[a:b].clone()), out[a:b]), (a, b)
        assert torch.equal(w.last_move_accepted, acc[a:b]) and torch.equal(w.last_e_after, ea[a:b])


def test_pair_distance_posterior_beyond_2_32_elements(device):
    """C5's model (256 beads, 768 coordinates per chain) with 5.6 million chains: the fused
    leapfrog on packed targets and the one-launch energy; the kernels' summation order does not
    depend on the batch, so windows agree bit for bit."""
    need(device, 125)
    from binf_amd.example.distance import make_distance_likelihood
    from binf_amd.pdf.posteriors import Posterior
    n = 256
    edge = (1 << 32) // (3 * n) + 1
    C = edge + 2048
    rs = np.random.RandomState(6)
    truth = rs.standard_normal((n, 3)) * 2.0
    d = truth[:, None, :] - truth[None, :, :]
    iu = np.triu_indices(n, 1)
    ys = np.sqrt((d ** 2).sum(-1))[iu]
    lik = make_distance_likelihood(ys, n)
    prior = IsotropicGaussian(0.05, 0.0, name='coordinates_prior', variable_name='coordinates')
    cond = Posterior({lik.name: lik}, {prior.name: prior}).conditional_factory(precision=4.0)
    g = torch.Generator(device=device).manual_seed(7)
    x = torch.randn((C, 3 * n), dtype=torch.float64, device=device, generator=g) * 0.1
    x += torch.from_numpy(truth.reshape(-1)).to(device)[None, :]
    p0 = torch.randn((C, 3 * n), dtype=torch.float64, device=device, generator=g)
    u = torch.rand(C, dtype=torch.float64, device=device, generator=g)
    s = HMCSampler(cond, x, 0.002, 1, variable_name='coordinates')
    out = s.sample(p0=p0, u=u)
    acc, eb, ea = s.last_move_accepted, s.last_e_before, s.last_e_after
    assert bool(torch.isfinite(ea).all()) and 0.5 < float(acc.double().mean()) <= 1.0
    for a, b in [(0, 300), (edge - 300, edge + 300), (C - 300, C)]:
        w = HMCSampler(cond, x[a:b].clone(), 0.002, 1, variable_name='coordinates')
        wout = w.sample(p0=p0[a:b].clone(), u=u[a:b].clone())
        assert torch.equal(wout, out[a:b]), (a, b)
        assert torch.equal(w.last_move_accepted, acc[a:b])
        assert torch.equal(w.last_e_before, eb[a:b]) and torch.equal(w.last_e_after, ea[a:b])

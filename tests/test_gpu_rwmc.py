"""The random-walk Metropolis subsampler and the conjugate Gamma draw ON THE
DEVICE: reference ``binf/example/samplers.py:43-51,78-92``.

The reference's own test of this wiring is ``example_script.py`` itself; the
bit-for-bit one-chain replay against the numpy restatement lives in
``tests/test_gpu_poly.py::test_example_script_itself_one_chain_same_stream``
(host np.random draws through the same two kernels).  Here: the device-draw
mode against the same kernels fed with the dumped draws, numpy's exp
semantics of the accept test, sharding, and that a sweep needs no host draw.
"""
import numpy as np
import pytest
import torch

from binf_amd import _native
from binf_amd.example.likelihood import POLYVAL
from binf_amd.example.misc import make_posterior
from binf_amd.example.samplers import GammaSampler, RWMCSampler, make_sampler
from binf_amd.samplers import BinfState
from binf_amd.samplers.rng import DeviceRNG
from oracle import ref_example as RE

pytestmark = pytest.mark.gpu


def dev_t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)


def uniform_dump(shape, seed, offset, chain_offset, device):
    out = torch.empty(shape, dtype=torch.float64, device=device)
    per_chain = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    _native.rng_fill('uniform', out, seed, offset, elem_offset=chain_offset * per_chain)
    return out


@pytest.mark.parametrize('C,K,coff', [(1, 4, 0), (33, 4, 0), (33, 5, 7), (1000, 33, 3), (70, 1, 1)])
def test_device_draws_equal_the_kernels_fed_with_their_dump(device, C, K, coff):
    """propose / accept with draws generated in the kernels == the same kernels
    fed with the Philox stream's dump (low + (high - low) * U formed as legacy
    numpy forms it), bit for bit; streams keyed by the global element / chain."""
    step, seed = 0.1, 77
    rs = np.random.RandomState(C * K)
    state = dev_t(rs.standard_normal((C, K)), device)
    prop_dev = _native.rwmc_propose(state, step, None, seed, 5, coff)
    U = uniform_dump((C, K), seed, 5, coff, device).cpu().numpy()
    change = -step + (step - -step) * U
    assert np.array_equal(prop_dev.cpu().numpy(), state.cpu().numpy() + change)
    prop_host = _native.rwmc_propose(state, step, dev_t(change, device))
    assert torch.equal(prop_dev, prop_host)
    lp_old = dev_t(rs.standard_normal(C), device)
    lp_new = dev_t(rs.standard_normal(C), device)
    outs = []
    for u in (None, uniform_dump((C,), seed, 6, coff, device)):
        out = torch.empty_like(state)
        acc = torch.empty(C, dtype=torch.uint8, device=device)
        nacc = torch.zeros(C, dtype=torch.int64, device=device)
        _native.rwmc_accept(prop_dev, state, lp_old, lp_new, out, acc, nacc, u, seed, 6, coff)
        outs.append((out, acc, nacc))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    out, acc, nacc = outs[0]
    uu = uniform_dump((C,), seed, 6, coff, device).cpu().numpy()
    want = uu < np.exp(-((-lp_new.cpu().numpy()) - (-lp_old.cpu().numpy())))
    assert np.array_equal(acc.cpu().numpy().astype(bool), want)
    assert np.array_equal(nacc.cpu().numpy(), want.astype(np.int64))
    assert torch.equal(out, torch.where(acc.bool()[:, None], prop_dev, state))
    # in place: state_out == proposal
    p2 = prop_dev.clone()
    _native.rwmc_accept(p2, state, lp_old, lp_new, p2, None, None, None, seed, 6, coff)
    assert torch.equal(p2, out)


def test_accept_uses_numpys_exp_not_the_clipped_one(device):
    """samplers.py:86 is np.exp, not csb's clipped exp (hmc.py:151): an energy
    drop beyond 709 overflows to inf (always accepted), a rise beyond 745
    underflows to exactly 0 (u = 0 is NOT accepted; the clipped exp(-308) would
    accept it), subnormal ratios survive, NaN rejects."""
    d = np.array([800.0, 710.0, 709.0, 0.0, -720.0, -745.0, -746.0, -1000.0, np.nan, np.inf, -np.inf])
    C = len(d)
    state = torch.zeros((C, 3), dtype=torch.float64, device=device)
    prop = torch.ones((C, 3), dtype=torch.float64, device=device)
    lp_old = torch.zeros(C, dtype=torch.float64, device=device)
    lp_new = dev_t(d, device)
    with np.errstate(over='ignore', invalid='ignore'):
        ratio = np.exp(d)
    for uval in (0.0, 5e-324, 1e-300, 0.999999, 0.5):
        u = torch.full((C,), uval, dtype=torch.float64, device=device)
        acc = torch.empty(C, dtype=torch.uint8, device=device)
        out = torch.empty_like(state)
        _native.rwmc_accept(prop, state, lp_old, lp_new, out, acc, None, u)
        want = uval < ratio
        got = acc.cpu().numpy().astype(bool)
        # the device exp may differ from numpy's in the last bit: exclude ties
        safe = ~np.isclose(ratio, uval, rtol=1e-14, atol=0.0) | (ratio == 0.0) | ~np.isfinite(ratio)
        assert np.array_equal(got[safe], want[safe]), (uval, got, want)
    assert ratio[0] == np.inf and ratio[6] == 0.0 and 0.0 < ratio[4] < 2.3e-308


def gibbs_run(device, C, sweeps, rng, coff=0, start=None):
    xs, ys = RE.example_data()
    st = BinfState(dict(
        coefficients=torch.ones((C, 4), dtype=torch.float64, device=device) if start is None else start[0],
        precision=torch.ones(C, dtype=torch.float64, device=device) if start is None else start[1]))
    gips = make_sampler(make_posterior(xs, ys, POLYVAL), 0.1, st, rng=rng)
    cs, ts = [], []
    def stream_pos():
        # key words AND position: a few draws move only the position
        st = np.random.get_state()
        return (st[1].tobytes(), st[2], st[3], st[4])
    before = stream_pos()
    for _ in range(sweeps):
        s = gips.sample()
        cs.append(s.variables['coefficients'].clone())
        ts.append(s.variables['precision'].clone())
    gips.host_stream_untouched = stream_pos() == before
    gips.host_stream_untouched_after = lambda _: stream_pos() == before
    return torch.stack(cs), torch.stack(ts), gips


def test_device_rwmc_gibbs_needs_no_host_draw_and_mixes(device):
    """make_sampler(rng=DeviceRNG): every draw of the sweep (uniform proposal,
    acceptance, gamma) is a device draw -- np.random is never consumed, the
    global stream stays where it was -- and the chains reach the posterior."""
    C, sweeps = 512, 400
    cs, ts, gips = gibbs_run(device, C, sweeps, DeviceRNG(9, device))
    assert gips.host_stream_untouched
    # a random walk of half-width 0.1 needs some 1e4 sweeps from the all-ones start:
    # one multi-sweep launch (GibbsSampler.sample_n), still without a host draw
    rec = gips.sample_n(40000, thin=200)
    assert gips.host_stream_untouched_after(rec)
    _, _, host = gibbs_run(device, 2, 2, None)
    assert not host.host_stream_untouched            # the parity mode does consume np.random
    rate = gips.last_draw_stats['coefficients'].acceptance_rate
    assert rate.shape == (C,) and 0.05 < float(rate.mean()) < 0.95
    assert isinstance(gips.subsamplers['precision'], GammaSampler)
    assert isinstance(gips.subsamplers['coefficients'], RWMCSampler)
    assert (ts > 0).all() and torch.isfinite(cs).all()
    # the posterior of the example: coefficients near the truth after burn-in
    m = rec['coefficients'][100:].mean(dim=(0, 1)).cpu().numpy()
    assert np.all(np.abs(m - np.array([2.0, -4.0, 1.0, 1.5])) < 1.0), m
    # deterministic in the seed
    cs2, ts2, _ = gibbs_run(device, C, 5, DeviceRNG(9, device))
    assert torch.equal(cs2, cs[:5]) and torch.equal(ts2, ts[:5])
    cs3, _, _ = gibbs_run(device, C, 5, DeviceRNG(10, device))
    assert not torch.equal(cs3, cs[:5])


@pytest.mark.parametrize('C,parts', [(64, 2), (37, 3)])
def test_sharded_rwmc_gibbs_reproduces_the_unsharded_run(device, C, parts):
    """RWMC + Gamma inside Gibbs with device draws: shards with chain offsets ==
    the full batch, bit for bit (proposal, acceptance and gamma streams are keyed
    by the global chain index)."""
    sweeps = 6
    full_c, full_t, _ = gibbs_run(device, C, sweeps, DeviceRNG(5, device))
    for r in range(parts):
        rng, start, count = DeviceRNG.for_shard(5, C, rank=r, world_size=parts, device=device)
        c, t, _ = gibbs_run(device, count, sweeps, rng)
        assert torch.equal(c, full_c[:, start:start + count]), r
        assert torch.equal(t, full_t[:, start:start + count]), r


def test_gamma_sampler_defaults_to_the_device_gamma_with_an_rng(device):
    xs, ys = RE.example_data()
    C = 16
    post = make_posterior(xs, ys, POLYVAL)
    coeffs = torch.ones((C, 4), dtype=torch.float64, device=device)
    pdf = post.conditional_factory(coefficients=coeffs)
    a = GammaSampler(pdf, None, rng=DeviceRNG(3, device))
    b = GammaSampler(pdf, None, gamma=DeviceRNG(3, device).gamma)
    assert torch.equal(a.sample(), b.sample())
    with pytest.raises(TypeError):
        GammaSampler(pdf, None, rng=object())

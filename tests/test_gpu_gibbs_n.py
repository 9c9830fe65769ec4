"""GPU parity tests of the multi-sweep Gibbs launch, ``GibbsSampler.sample_n`` /
``binf_gibbs_poly_sample_n_f64`` (csrc/gibbs_poly.hip): the loop
``for i in range(n): gips.sample()`` of the reference's ``example_script.py:33-34``
around ``binf/samplers/gibbs.py:136-151`` in ONE launch.

Bars: BIT-IDENTICAL to n single sweeps (same draws: the in-kernel Philox streams
are the streams the stand-alone generator kernels write; host draws are consumed
in the reference's order), and -- through the single sweeps and directly --
against the numpy restatement of the example (oracle/ref_example.py): bit for bit
for the reference's own RWMC + Gamma script, inside the computed 1e-10 bounds of
tests/poly_bounds.py for the HMC wiring (the force is a BLAS-order contraction in
the reference)."""
import numpy as np
import pytest
import torch

from binf_amd import _native, native
from binf_amd.example import native_poly
from binf_amd.example.likelihood import POLYVAL, ForwardModel, GaussianErrorModel
from binf_amd.example.misc import make_posterior
from binf_amd.example.priors import GammaPrior, GaussianPrior
from binf_amd.example.samplers import make_hmc_sampler, make_sampler
from binf_amd.pdf.likelihoods import Likelihood
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers import BinfState
from binf_amd.samplers.rng import DeviceRNG, HostLegacyRNG
import poly_bounds as PB
from conftest import golden_files, load_golden
from oracle import ref_example as RE
from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu


def dev_t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)


def model(K, N, seed, xlim=1.5):
    rs = np.random.RandomState(seed)
    xs = np.linspace(-xlim, xlim, N)
    c_true = rs.standard_normal(K)
    ys = R.polyval(xs, c_true) + rs.standard_normal(N) / np.sqrt(2.5)
    return xs, ys, c_true


def posterior(xs, ys, K):
    lik = Likelihood('points', ForwardModel(xs, POLYVAL), GaussianErrorModel(ys))
    return Posterior({lik.name: lik}, {'precision_prior': GammaPrior(1.0, 0.2),
                                       'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K) * 5)})


def build(device, move, K, N, C, seed, rng, dt=None, L=7, start=None, **kw):
    xs, ys, c_true = model(K, N, seed)
    rs = np.random.RandomState(seed + 1)
    if start is None:
        start = (dev_t(c_true + 0.2 * rs.standard_normal((C, K)), device),
                 dev_t(1.0 + rs.uniform(size=C), device))
    st = BinfState(dict(coefficients=start[0].clone(), precision=start[1].clone()))
    post = posterior(xs, ys, K)
    if move == 'hmc':
        dt = dt if dt is not None else 0.02 / (K * np.sqrt(N / 20.0))
        hk = dict(kw)
        if rng is not None:
            hk['rng'] = rng
        return make_hmc_sampler(post, dt, L, st, record_energies=True, **hk)
    return make_sampler(post, 0.05, st, rng=rng)


def run_loop(gips, n):
    # the per-variable sweep of gibbs.py:136-151 (subsampler by subsampler: the separate
    # transition, chi^2, generator and update kernels), NOT the one-launch sweep
    gips.fused_sweep = False
    cs, ts, fl = [], [], []
    for _ in range(n):
        s = gips.sample()
        cs.append(s.variables['coefficients'].clone())
        ts.append(s.variables['precision'].clone())
        fl.append(gips.subsamplers['coefficients'].last_move_accepted.clone())
    return torch.stack(cs), torch.stack(ts), torch.stack(fl)


SHAPES = [(4, 20, 37), (4, 20, 512), (7, 37, 9), (16, 128, 21), (3, 7, 70), (8, 300, 19),
          (5, 129, 8), (16, 1024, 5), (1, 1, 3), (15, 100, 11)]


@pytest.mark.parametrize('move', ['hmc', 'rwmc'])
@pytest.mark.parametrize('K,N,C', SHAPES)
def test_sample_n_is_n_sweeps_bit_for_bit_with_device_draws(device, move, K, N, C):
    n, thin = 6, 2
    if N == 1:
        pytest.skip('gamma shape 0.5 N < 1: generated variates not covered, see the fallback test')
    a = build(device, move, K, N, C, 11, DeviceRNG(5, device))
    b = build(device, move, K, N, C, 11, DeviceRNG(5, device))
    cs, ts, fl = run_loop(a, n)
    rec = b.sample_n(n, thin=thin)
    assert torch.equal(rec['coefficients'], cs[thin - 1::thin])
    assert torch.equal(rec['precision'], ts[thin - 1::thin])
    sa, sb = a.subsamplers['coefficients'], b.subsamplers['coefficients']
    assert torch.equal(b.state.variables['coefficients'], cs[-1])
    assert torch.equal(b.state.variables['precision'], ts[-1])
    assert torch.equal(sb.last_move_accepted, fl[-1])
    if move == 'hmc':
        assert torch.equal(sb.accepted_history, fl)
        assert torch.equal(sb.n_accepted, sa.n_accepted) and sb.counter == sa.counter == n
        assert torch.equal(sb.last_e_before, sa.last_e_before)
        assert torch.equal(sb.last_e_after, sa.last_e_after)
        assert sb.rng.offset == sa.rng.offset
    else:
        assert torch.equal(sb._n_accepted_moves, sa._n_accepted_moves) and sb._n_moves == n
        assert torch.equal(sb.acceptance_rate, sa.acceptance_rate)
        assert sb.rng.offset == sa.rng.offset
    # something moved, and the two loops stay in step afterwards
    assert 0 < int(fl.sum()) or move == 'rwmc'
    x, y = a.sample(), b.sample()            # per-variable sweep vs the sweep as one launch
    assert a.fused_sweep is False and b.fused_sweep is True
    assert torch.equal(x.variables['coefficients'], y.variables['coefficients'])
    assert torch.equal(x.variables['precision'], y.variables['precision'])
    assert sa.rng.offset == sb.rng.offset


@pytest.mark.parametrize('mode', ['exact', 'fma'])
def test_sample_n_hmc_with_adaption_and_both_arithmetic_modes(device, mode):
    K, N, C, n = 4, 20, 100, 9
    kw = dict(timestep_adaption_limit=6, mode=mode)
    a = build(device, 'hmc', K, N, C, 3, DeviceRNG(1, device), dt=0.03, **kw)
    b = build(device, 'hmc', K, N, C, 3, DeviceRNG(1, device), dt=0.03, **kw)
    cs, ts, fl = run_loop(a, n)
    rec = b.sample_n(4)
    rec2 = b.sample_n(n - 4)                       # the adaption window ends inside a launch
    assert torch.equal(torch.cat([rec['coefficients'], rec2['coefficients']]), cs)
    assert torch.equal(torch.cat([rec['precision'], rec2['precision']]), ts)
    sa, sb = a.subsamplers['coefficients'], b.subsamplers['coefficients']
    assert torch.equal(sb.timestep, sa.timestep) and sb.timestep.shape == (C,)
    assert len(torch.unique(sb.timestep)) > 1
    assert torch.equal(sb.n_accepted, sa.n_accepted)


@pytest.mark.parametrize('two_generators', [False, True])
def test_draw_streams_follow_the_generators_of_the_subsamplers(device, two_generators):
    """One DeviceRNG serving every draw (make_hmc_sampler's default) or one for
    the HMC draws and another for the gamma variates: the launch reserves the
    stream positions n single sweeps would take."""
    K, N, C, n = 4, 20, 33, 5

    def mk():
        h = DeviceRNG(7, device, normal='box_muller' if two_generators else 'ziggurat')
        if two_generators:
            return build(device, 'hmc', K, N, C, 2, h, gamma=DeviceRNG(8, device).gamma), h
        return build(device, 'hmc', K, N, C, 2, h), h
    (a, ha), (b, hb) = mk(), mk()
    cs, ts, _ = run_loop(a, n)
    rec = b.sample_n(n)
    assert torch.equal(rec['coefficients'], cs) and torch.equal(rec['precision'], ts)
    assert ha.offset == hb.offset
    ga = getattr(a.subsamplers['precision'].gamma, '__self__')
    gb = getattr(b.subsamplers['precision'].gamma, '__self__')
    assert ga.offset == gb.offset and (ga is not ha) == two_generators


@pytest.mark.parametrize('move', ['hmc', 'rwmc'])
def test_sample_n_with_the_host_stream_consumes_it_like_n_sweeps(device, move):
    K, N, C, n = 4, 20, 5, 8
    a = build(device, move, K, N, C, 4, HostLegacyRNG() if move == 'hmc' else None)
    b = build(device, move, K, N, C, 4, HostLegacyRNG() if move == 'hmc' else None)
    np.random.seed(77)
    cs, ts, _ = run_loop(a, n)
    end_a = np.random.get_state()
    np.random.seed(77)
    rec = b.sample_n(n)
    end_b = np.random.get_state()
    assert torch.equal(rec['coefficients'], cs) and torch.equal(rec['precision'], ts)
    assert end_a[2] == end_b[2] and np.array_equal(end_a[1], end_b[1])


@pytest.mark.parametrize('seed', [0, 1, 7])
def test_example_script_itself_in_one_launch(device, seed):
    """example_script.py as the reference ships it (ONE chain, RWMC + Gamma inside
    Gibbs, every draw from the global np.random stream that also made the data),
    300 sweeps in ONE launch: every state equals the numpy restatement
    oracle/ref_example.py:example_script_chain bit for bit."""
    sweeps = 300
    ref = RE.example_script_chain(seed, sweeps)
    np.random.seed(seed)
    xs = np.linspace(-2, 2, 20)
    ys = np.random.normal(loc=R.polyval(xs, np.array([2.0, -4.0, 1.0, 1.5])),
                          scale=1.0 / np.sqrt(2.5))
    start = BinfState(dict(coefficients=dev_t(np.ones((1, 4)), device),
                           precision=dev_t(np.ones(1), device)))
    gips = make_sampler(make_posterior(xs, ys, POLYVAL), 0.1, start)
    rec = gips.sample_n(sweeps)
    assert np.array_equal(rec['coefficients'].cpu().numpy()[:, 0], ref['coefficients'])
    assert np.array_equal(rec['precision'].cpu().numpy()[:, 0], ref['precision'])
    rate = gips.last_draw_stats['coefficients'].acceptance_rate
    assert abs(float(rate) - ref['acceptance_rate']) < 1e-12


@pytest.mark.parametrize('path', golden_files('poly_'))
def test_one_launch_reproduces_the_golden_gibbs_within_hmc_vectors(device, path):
    """tests/golden/poly_*.npz (Gibbs-within-HMC sweeps of the restatement with the
    draws recorded in the reference's consumption order) through the C ABI entry
    point directly, all sweeps in one launch, draws supplied."""
    g = load_golden(path)
    K, N, L, dt = int(g['K']), int(g['N']), int(g['L']), float(g['timestep'])
    if K > 16 or N > 1024:
        with pytest.raises(NotImplementedError):
            _run_golden(g, device)
        return
    out = _run_golden(g, device)
    S, C = g['u'].shape
    pb = PB.PolyBound(g['xs'], g['ys'], K, np.zeros(K), np.ones(K) * 5)
    assert np.array_equal(out['acc'].cpu().numpy(), g['accepted'].astype(np.uint8))
    c, t = out['rc'].cpu().numpy(), out['rt'].cpu().numpy()
    eb, ea = out['eb'].cpu().numpy(), out['ea'].cpu().numpy()
    for k in range(C):
        b = PB.gibbs_bounds(pb, g['coefficients'][:, k], g['accepted'][:, k], g['p0'][:, k],
                            g['precision'][:, k], g['precision0'][k], g['coefficients0'][k],
                            dt, L, RE.PRIOR_RATE_IN_CONDITIONALS)
        for s in range(S):
            want = g['coefficients'][s][k]
            assert np.all(np.abs(c[s, k] - want) <= b[s]['bq'] + 4 * PB.U * np.abs(want)), (s, k)
            assert abs(t[s, k] - g['precision'][s][k]) <= b[s]['btau'] * g['precision'][s][k], (s, k)
            assert abs(eb[s, k] - g['e_before'][s][k]) <= b[s]['be_before'], (s, k)
            assert abs(ea[s, k] - g['e_after'][s][k]) <= b[s]['be_after'], (s, k)


def _run_golden(g, device):
    K, N, L, dt = int(g['K']), int(g['N']), int(g['L']), float(g['timestep'])
    S, C = g['u'].shape
    th, tau = dev_t(g['coefficients0'], device), dev_t(g['precision0'], device)
    o = dict(rc=torch.empty((S, C, K), dtype=torch.float64, device=device),
             rt=torch.empty((S, C), dtype=torch.float64, device=device),
             acc=torch.empty((S, C), dtype=torch.uint8, device=device),
             eb=torch.empty((S, C), dtype=torch.float64, device=device),
             ea=torch.empty((S, C), dtype=torch.float64, device=device))
    _native.gibbs_poly_sample_n(
        th, tau, torch.empty_like(th), torch.empty_like(tau), dev_t(g['xs'], device),
        dev_t(g['ys'], device), S, 1, move=_native.MOVE_HMC, nsteps=L, timestep=dt,
        prior_means=dev_t(np.zeros(K), device), prior_vars=dev_t(np.ones(K) * 5, device),
        prior_first=True, gp_where=2, gp_shape=1.0, gp_rate=RE.PRIOR_RATE_IN_CONDITIONALS,
        gamma_shape=float(g['gamma_shape']), gamma_rate=RE.PRIOR_RATE_IN_CONDITIONALS,
        rec_coefficients=o['rc'], rec_precision=o['rt'], accepted=o['acc'], e_before=o['eb'],
        e_after=o['ea'], p0=dev_t(g['p0'], device), u=dev_t(g['u'], device),
        g=dev_t(g['gamma'], device))
    torch.cuda.synchronize()
    return o


@pytest.mark.parametrize('move', ['hmc', 'rwmc'])
@pytest.mark.parametrize('C,parts', [(64, 2), (37, 3)])
def test_sharded_launches_reproduce_the_unsharded_one(device, move, C, parts):
    K, N, n = 4, 20, 5
    full = build(device, move, K, N, C, 9, DeviceRNG(3, device))
    start = (full.state.variables['coefficients'].clone(), full.state.variables['precision'].clone())
    rec = full.sample_n(n)
    for r in range(parts):
        rng, s0, cnt = DeviceRNG.for_shard(3, C, rank=r, world_size=parts, device=device)
        part = build(device, move, K, N, cnt, 9, rng,
                     start=(start[0][s0:s0 + cnt], start[1][s0:s0 + cnt]))
        pr = part.sample_n(n)
        assert torch.equal(pr['coefficients'], rec['coefficients'][:, s0:s0 + cnt]), r
        assert torch.equal(pr['precision'], rec['precision'][:, s0:s0 + cnt]), r


def test_other_schemes_fall_back_to_the_loop(device):
    """A scheme the fused launch does not cover -- here a generated gamma shape
    below 1 (one data point) and a user callable for the gamma variates -- is
    swept one sample() at a time, same results as the caller's own loop."""
    a = build(device, 'hmc', 2, 1, 6, 5, DeviceRNG(2, device))
    b = build(device, 'hmc', 2, 1, 6, 5, DeviceRNG(2, device))
    cs, ts, _ = run_loop(a, 4)
    rec = b.sample_n(4, thin=2)
    assert torch.equal(rec['coefficients'], cs[1::2]) and torch.equal(rec['precision'], ts[1::2])
    calls = []

    def user_gamma(shape, n, dev):
        calls.append(shape)
        return torch.full((n,), shape, dtype=torch.float64, device=dev)
    c = build(device, 'hmc', 4, 20, 6, 5, DeviceRNG(2, device), gamma=user_gamma)
    assert c.sample_n(3, record=False) is None and len(calls) == 3


def test_entry_point_argument_checks(device):
    K, N, C = 4, 20, 8
    xs, ys, _ = model(K, N, 0)
    th = torch.zeros((C, K), dtype=torch.float64, device=device)
    tau = torch.ones(C, dtype=torch.float64, device=device)
    base = dict(move=_native.MOVE_RWMC, stepsize=0.1, gamma_shape=10.0, gamma_rate=1.0,
                streams=((1, 0, 130), (1, 1, 130), (1, 2, 130)))
    xs_d, ys_d = dev_t(xs, device), dev_t(ys, device)
    _native.gibbs_poly_sample_n(th, tau, th.clone(), tau.clone(), xs_d, ys_d, 2, **base)
    with pytest.raises(ValueError):
        _native.gibbs_poly_sample_n(th, tau, th.clone(), tau.clone(), xs_d, ys_d, 0, **base)
    with pytest.raises(ValueError):
        _native.gibbs_poly_sample_n(th, tau, th.clone(), tau.clone(), xs_d, ys_d, 2,
                                    **dict(base, gp_where=3))
    with pytest.raises(NotImplementedError):
        _native.gibbs_poly_sample_n(th, tau, th.clone(), tau.clone(), xs_d, ys_d, 2,
                                    **dict(base, gamma_shape=0.5))
    with pytest.raises(NotImplementedError):
        big = torch.zeros((C, 17), dtype=torch.float64, device=device)
        _native.gibbs_poly_sample_n(big, tau, big.clone(), tau.clone(), xs_d, ys_d, 2, **base)
    with pytest.raises(ValueError):          # partial overlap of state and output
        buf = torch.zeros(C * K + K, dtype=torch.float64, device=device)
        _native.gibbs_poly_sample_n(buf[:C * K].view(C, K), tau, buf[K:].view(C, K), tau.clone(),
                                    xs_d, ys_d, 2, **base)
    # in place, and no chains at all
    t2, p2 = th.clone(), tau.clone()
    _native.gibbs_poly_sample_n(t2, p2, t2, p2, xs_d, ys_d, 3, **base)
    ref_t, ref_p = th.clone(), tau.clone()
    _native.gibbs_poly_sample_n(th, tau, ref_t, ref_p, xs_d, ys_d, 3, **base)
    assert torch.equal(t2, ref_t) and torch.equal(p2, ref_p)
    e = torch.zeros((0, K), dtype=torch.float64, device=device)
    _native.gibbs_poly_sample_n(e, e[:, 0].contiguous(), e.clone(), e[:, 0].contiguous(),
                                xs_d, ys_d, 2, **base)
    torch.cuda.synchronize()


@pytest.mark.parametrize('K,N,C,source,per_chain_tau', [
    (4, 20, 37, 'device', True), (7, 300, 9, 'device', False), (16, 128, 5, 'supplied', True),
    (4, 20, 3, 'host', False), (3, 1000, 4, 'device', True)])
def test_hmc_sample_n_on_the_coefficient_conditional_is_n_sample_calls(device, K, N, C, source,
                                                                       per_chain_tau):
    """HMCSampler.sample_n on the example's conditional posterior of the coefficients
    (precision fixed): the multi-sweep launch with its precision draw switched off,
    bit for bit n sample() calls -- states, flags, energies, adapted step sizes,
    generator positions."""
    from binf_amd.samplers.hmc import HMCSampler
    n, thin, L = 7, 3, 5
    xs, ys, c_true = model(K, N, 21)
    rs = np.random.RandomState(K + N)
    tau = dev_t(1.0 + rs.uniform(size=C), device) if per_chain_tau else 2.5
    cond = posterior(xs, ys, K).conditional_factory(precision=tau)
    theta = dev_t(c_true + 0.2 * rs.standard_normal((C, K)), device)
    dt = 0.02 / (K * np.sqrt(N / 20.0))
    p0 = dev_t(rs.standard_normal((n, C, K)), device)
    u = dev_t(rs.uniform(size=(n, C)), device)

    def mk():
        rng = DeviceRNG(4, device) if source == 'device' else HostLegacyRNG()
        return HMCSampler(cond, theta.clone(), dt, L, variable_name='coefficients', rng=rng,
                          timestep_adaption_limit=5, record_energies=True)
    a, b = mk(), mk()
    np.random.seed(5)
    xs_loop = [a.sample(p0=p0[i], u=u[i]) if source == 'supplied' else a.sample() for i in range(n)]
    flags = torch.stack([a_.clone() for a_ in [a.last_move_accepted]])
    np.random.seed(5)
    rec = b.sample_n(n, thin=thin, **(dict(p0=p0, u=u) if source == 'supplied' else {}))
    assert rec.shape == (n // thin, C, K)
    assert torch.equal(rec[0], xs_loop[thin - 1]) and torch.equal(rec[1], xs_loop[2 * thin - 1])
    assert torch.equal(b.state, xs_loop[-1]) and b.counter == a.counter == n
    assert torch.equal(b.n_accepted, a.n_accepted)
    assert torch.equal(b.last_move_accepted, flags[0])
    assert torch.equal(b.timestep, a.timestep)
    assert torch.equal(b.last_e_after[-1], a.last_e_after)
    if source == 'device':
        assert a.rng.offset == b.rng.offset == 2 * n
    # ... and they stay in step
    if source != 'supplied':
        np.random.seed(6)
        x = a.sample()
        np.random.seed(6)
        assert torch.equal(x, b.sample())


def test_layout_of_the_fused_transition_follows_the_batch_consistently(device):
    """Up to LANE_MIN_CHAINS chains a chain's data are spread over a lane group,
    from there on (<= 128 data points) a chain is one lane -- for sample(), for the
    one-launch sweep and for sample_n alike, so the three agree bit for bit at any
    batch size; the two layouts differ at rounding level only (the force's order)."""
    assert native_poly.LANE_MIN_CHAINS == 65536
    old = native_poly.LANE_MIN_CHAINS
    native_poly.LANE_MIN_CHAINS = 64              # the rule with a threshold a test can afford
    try:
        K, N, n = 4, 20, 4
        for C in (63, 64):
            a = build(device, 'hmc', K, N, C, 1, DeviceRNG(2, device))
            b = build(device, 'hmc', K, N, C, 1, DeviceRNG(2, device))
            c = build(device, 'hmc', K, N, C, 1, DeviceRNG(2, device))
            spec = a.subsamplers['coefficients'].pdf.native_hmc_spec('coefficients')
            assert native.get(spec).extras['lane_layout'](a.subsamplers['coefficients'], spec, C) is (C >= 64)
            cs, ts, _ = run_loop(a, n)                      # per-variable sweeps
            rec = b.sample_n(n)                             # one call
            for _ in range(n):
                c.sample()                                  # default sweeps
            assert torch.equal(rec['coefficients'], cs) and torch.equal(rec['precision'], ts)
            assert torch.equal(c.state.variables['coefficients'], cs[-1])
        # the same chains under the two layouts: equal to 1e-12, not to the bit
        g = build(device, 'hmc', K, N, 64, 1, DeviceRNG(2, device))
        g.subsamplers['coefficients'].fused_transition = 'group'
        x = g.sample_n(n)['coefficients']
        assert not torch.equal(x, cs) and torch.allclose(x, cs, rtol=1e-10, atol=1e-12)
    finally:
        native_poly.LANE_MIN_CHAINS = old

"""GPU parity tests of the pairwise-distance-restraint posterior (BASELINE
config C5).  The model is BUILD-DEFINED (the reference has no code for it):
parity is against the numpy formulation in oracle/ref_distance.py only --
"parity unpinned by the reference".  Distances and chi^2 are bit-exact; the
all-pairs force is held to 1e-10 (different summation order than np.add.at)."""
import numpy as np
import pytest
import torch

from binf_amd import _native
from binf_amd.example.distance import (DistanceErrorModel, DistanceForwardModel,
                                       make_distance_likelihood)
from binf_amd.pdf import IsotropicGaussian
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
from conftest import golden_files, load_golden
from oracle import ref_distance as RD
from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu


def dev_t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)


def synth(n, C, seed):
    rs = np.random.RandomState(seed)
    truth = rs.standard_normal((n, 3)) * 2.0
    ys = RD.forward(truth.reshape(-1), n) + 0.05 * rs.standard_normal(n * (n - 1) // 2)
    ys = np.abs(ys)
    x = truth.reshape(-1)[None, :] + 0.3 * rs.standard_normal((C, 3 * n))
    return ys, x


@pytest.mark.parametrize('n,C', [(2, 3), (3, 5), (10, 17), (64, 9), (256, 6), (300, 2)])
def test_forward_distances_bitwise(device, n, C):
    ys, x = synth(n, C, n)
    fwm = DistanceForwardModel(n)
    got = fwm(coordinates=dev_t(x, device)).cpu().numpy()
    want = np.stack([RD.forward(x[c], n) for c in range(C)])
    assert got.shape == (C, n * (n - 1) // 2)
    assert np.array_equal(got, want)


@pytest.mark.parametrize('n,C', [(3, 5), (10, 17), (64, 9), (256, 6), (300, 2),
                                 # around the 64-bead blocks of the n <= 256 force scheme
                                 # (32 .. 256 beads; below and above, the one-sided loops)
                                 (2, 2), (31, 3), (32, 3), (33, 3), (63, 3), (65, 3), (127, 2), (128, 2),
                                 (129, 2), (191, 2), (192, 2), (193, 2), (255, 2), (257, 2)])
def test_likelihood_logp_and_force(device, n, C):
    ys, x = synth(n, C, n + 1)
    L = make_distance_likelihood(ys, n)
    assert L.variables == {'coordinates', 'precision'}
    tx = dev_t(x, device)
    lp1 = L.log_prob(coordinates=tx, precision=1.0).cpu().numpy()
    assert np.array_equal(lp1, np.array([RD.log_prob(x[c], ys, 1.0, n) for c in range(C)]))
    taus = np.random.RandomState(0).uniform(0.5, 3.0, size=C)
    lp = L.log_prob(coordinates=tx, precision=dev_t(taus, device)).cpu().numpy()
    want = np.array([RD.log_prob(x[c], ys, taus[c], n) for c in range(C)])
    assert np.allclose(lp, want, rtol=1e-13)
    for prec, tarr in ((2.5, np.full(C, 2.5)), (dev_t(taus, device), taus)):
        g = L.gradient(coordinates=tx, precision=prec).cpu().numpy()
        for c in range(C):
            w = RD.gradient(x[c], ys, tarr[c], n)
            assert np.abs(g[c] - w).max() <= 1e-10 * max(np.abs(w).max(), 1.0), (n, c)
    # translation invariance: the restraint force sums to zero over the beads
    g = L.gradient(coordinates=tx, precision=2.5).cpu().numpy().reshape(C, n, 3)
    assert np.abs(g.sum(axis=1)).max() <= 1e-9 * np.abs(g).max()


def make_post(ys, n, k=0.05):
    L = make_distance_likelihood(ys, n)
    prior = IsotropicGaussian(k, 0.0, name='coordinates_prior', variable_name='coordinates')
    return Posterior({L.name: L}, {prior.name: prior})


@pytest.mark.parametrize('n,C,L,dt', [(8, 12, 10, 0.02), (256, 3, 4, 0.002)])
def test_hmc_on_distance_posterior_vs_restatement(device, n, C, L, dt):
    ys, x = synth(n, C, 3 * n)
    rs = np.random.RandomState(n)
    p0 = rs.standard_normal((C, 3 * n))
    u = rs.uniform(size=C)
    cond = make_post(ys, n).conditional_factory(precision=4.0)
    assert cond.variables == {'coordinates'}
    s = HMCSampler(cond, dev_t(x, device), dt, L, variable_name='coordinates')
    out = s.sample(p0=dev_t(p0, device), u=dev_t(u, device)).cpu().numpy()
    acc = s.last_move_accepted.cpu().numpy()
    for c in range(C):
        ref = R.RefHMCSampler(RD.DistancePosterior(ys, 4.0, n, prior_k=0.05), x[c].copy(),
                              dt, L, variable_name='coordinates',
                              normal=lambda size, c=c: p0[c].copy(), uniform=lambda c=c: u[c])
        want = ref.sample()
        assert bool(acc[c]) == bool(ref.last_move_accepted)
        assert np.abs(out[c] - want).max() <= 1e-9 * np.abs(want).max()
        assert abs(float(s.last_e_before[c]) - ref.last_E_before) <= 1e-10 * abs(ref.last_E_before)


def test_c5_size_properties(device):
    """BASELINE C5 per-GPU share: 3 x 256 coordinates, 256 chains (2048 / 8)."""
    n, C = 256, 256
    ys, x = synth(n, C, 99)
    L = make_distance_likelihood(ys, n)
    tx = dev_t(x, device)
    g = L.gradient(coordinates=tx, precision=2.0)
    torch.cuda.synchronize()
    gn = g.cpu().numpy().reshape(C, n, 3)
    assert np.abs(gn.sum(axis=1)).max() <= 1e-9 * np.abs(gn).max()
    # rotation invariance of the log-prob (distances only)
    th = 0.7
    Rm = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    xr = (x.reshape(C, n, 3) @ Rm.T).reshape(C, -1)
    a = L.log_prob(coordinates=tx, precision=2.0).cpu().numpy()
    b = L.log_prob(coordinates=dev_t(xr, device), precision=2.0).cpu().numpy()
    assert np.allclose(a, b, rtol=1e-11)
    for c in (0, 255):
        w = RD.gradient(x[c], ys, 2.0, n)
        assert np.abs(g[c].cpu().numpy() - w).max() <= 1e-10 * np.abs(w).max()


@pytest.mark.parametrize('n,C,L,dt,with_prior', [(8, 12, 10, 0.02, True), (256, 5, 6, 0.002, True),
                                                 (300, 3, 3, 0.002, False), (700, 2, 2, 0.001, True),
                                                 (1000, 2, 2, 0.0005, True), (770, 1030, 1, 0.0005, False)])
def test_fused_leapfrog_is_bit_identical_to_the_per_step_tier(device, n, C, L, dt, with_prior):
    ys, x = synth(n, C, 5 * n)
    rs = np.random.RandomState(n + 1)
    p0 = rs.standard_normal((C, 3 * n))
    u = rs.uniform(size=C)
    taus = rs.uniform(1.0, 5.0, size=C)
    outs = []
    for fused in (True, False):
        if with_prior:
            post = make_post(ys, n)
        else:
            L_ = make_distance_likelihood(ys, n)
            post = Posterior({L_.name: L_}, {})
        cond = post.conditional_factory(precision=dev_t(taus, device))
        spec = cond.native_leapfrog_spec('coordinates')
        assert spec is not None and spec[0] == 'pairdist' and (spec[3] is not None) == with_prior
        s = HMCSampler(cond, dev_t(x, device), dt, L, variable_name='coordinates')
        s.fused_leapfrog = fused
        out = s.sample(p0=dev_t(p0, device), u=dev_t(u, device))
        outs.append((out.cpu().numpy(), s.last_move_accepted.cpu().numpy(),
                     s.last_e_after.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2], outs[1][2])


def test_leapfrog_spec_rejects_unsupported_structures(device):
    ys, x = synth(8, 2, 1)
    # precision still free -> no fused leapfrog
    assert make_post(ys, 8).native_leapfrog_spec('coordinates') is None
    # a polynomial posterior has its own (binf_poly_leapfrog_f64), never the pair-distance one
    from binf_amd.example.misc import make_posterior
    from binf_amd.example.likelihood import POLYVAL
    post = make_posterior(np.linspace(-1, 1, 5), np.zeros(5), POLYVAL)
    spec = post.conditional_factory(precision=1.0).native_leapfrog_spec('coefficients')
    assert spec is not None and spec[0] == 'poly'
    assert post.native_leapfrog_spec('coefficients') is None          # precision still free


@pytest.mark.parametrize('n', [24, 40, 64, 100, 150, 200, 256, 300])
def test_force_is_independent_of_the_batch_size(device, n):
    """A chain's force and trajectory do not depend on how many other chains
    share the launch: from 32 to 256 beads every batch size runs the same
    one-workgroup-per-chain scheme (1, 4, 9 or 16 waves by the bead count; a
    summation order fixed by n alone, workgroups walking several chains when
    there are more chains than the chip holds); outside that range, few chains
    use four lanes per bead and many chains one, and both add in the same order."""
    ys, x = synth(n, 1100 if n > 64 else 5000, 7)     # more chains than workgroup slots on the chip
    L_ = make_distance_likelihood(ys, n)
    big = L_.gradient(coordinates=dev_t(x, device), precision=2.0).cpu().numpy()
    small = L_.gradient(coordinates=dev_t(x[:40], device), precision=2.0).cpu().numpy()
    assert np.array_equal(big[:40], small)
    q1, p1 = dev_t(x, device), dev_t(x[::-1].copy(), device)
    q2, p2 = q1[:40].clone(), p1[:40].clone()
    em = L_.error_model
    _native.pairdist_leapfrog(q1, p1, em.ymat_device(device), 2.0, (0.05, 0.0), True, 0.002, None, 4)
    _native.pairdist_leapfrog(q2, p2, em.ymat_device(device), 2.0, (0.05, 0.0), True, 0.002, None, 4)
    assert torch.equal(q1[:40], q2) and torch.equal(p1[:40], p2)


def test_force_counts_every_pair_once_in_each_direction(device):
    """The 32 <= n <= 256 scheme evaluates an unordered pair once and books it on both
    beads.  With all targets 0 the pair weight is exactly 1, so the force on
    bead i is sum_j (x_i - x_j): integer coordinates make every partial sum
    exact, and any pair dropped, doubled or booked on the wrong bead shows."""
    for n in (2, 31, 32, 33, 64, 65, 128, 130, 192, 200, 256):
        rs = np.random.RandomState(n)
        x = rs.randint(-50, 50, size=(3, n, 3)).astype(np.float64)
        # distinct beads (a zero distance has no direction)
        x[:, :, 0] += 200.0 * np.arange(n)[None, :]
        ymat = torch.zeros((n, n), dtype=torch.float64, device=device)
        g = _native.pairdist_gauss_grad(dev_t(x.reshape(3, -1), device), ymat, 1.0).cpu().numpy()
        want = n * x - x.sum(axis=1, keepdims=True)
        assert np.array_equal(g.reshape(3, n, 3), want), n


@pytest.mark.parametrize('n,C', [(2, 3), (3, 2), (9, 4), (17, 5), (128, 4), (129, 3), (256, 6), (300, 2),
                                 (700, 2), (17, 1030), (130, 1025), (17, 2049), (100, 2051), (131, 2048)])
def test_fused_log_prob_is_forward_plus_error_model_bitwise(device, n, C):
    """binf_pairdist_gauss_logp_f64 (distances never written to HBM) against
    the two-kernel path it replaces, including n_pairs = 8128 / 8256 / 32640 /
    44850 (one chunk, ragged second chunk, several chunks of the np.sum order) and
    chain counts from 2048 up, where one workgroup sums two chains over one pass of
    the pair list (an odd count: the last workgroup has one chain)."""
    ys, x = synth(n, C, 3 * n)
    L = make_distance_likelihood(ys, n)
    tx = dev_t(x, device)
    taus = dev_t(np.random.RandomState(n).uniform(0.5, 3.0, size=C), device)
    for prec in (1.0, 2.5, taus):
        fused = L.log_prob(coordinates=tx, precision=prec)
        mock = L.forward_model(coordinates=tx)
        two = _native.gauss_err_logp(mock, L.error_model.ys_device(device), prec)
        assert torch.equal(fused, two)


@pytest.mark.parametrize('path', golden_files('dist_'))
def test_distance_posterior_reproduces_golden_vectors(device, path):
    """Committed fixtures (tests/golden/dist_*.npz): likelihood log-prob bit for
    bit, force to 1e-10, consecutive HMC transitions with the recorded draws."""
    g = load_golden(path)
    n, L, dt = int(g['n_beads']), int(g['L']), float(g['timestep'])
    tau, pk = float(g['precision']), float(g['prior_k'])
    lik = make_distance_likelihood(g['ys'], n)
    x0 = dev_t(g['q0'], device)
    assert np.array_equal(lik.log_prob(coordinates=x0, precision=tau).cpu().numpy(),
                          g['likelihood_log_prob'])
    gr = lik.gradient(coordinates=x0, precision=tau).cpu().numpy()
    assert np.abs(gr - g['likelihood_gradient']).max() <= 1e-10 * np.abs(g['likelihood_gradient']).max()
    priors = {}
    if pk != 0.0:
        priors['coordinates_prior'] = IsotropicGaussian(pk, 0.0, name='coordinates_prior',
                                                        variable_name='coordinates')
    cond = Posterior({lik.name: lik}, priors).conditional_factory(precision=tau)
    s = HMCSampler(cond, x0, dt, L, variable_name='coordinates')
    for i in range(g['u'].shape[0]):
        out = s.sample(p0=dev_t(g['p0'][i], device), u=dev_t(g['u'][i], device)).cpu().numpy()
        assert np.array_equal(s.last_move_accepted.cpu().numpy(), g['accepted'][i].astype(bool))
        assert np.abs(out - g['q_out'][i]).max() <= 1e-9 * np.abs(g['q_out'][i]).max()
        assert np.allclose(s.last_e_before.cpu().numpy(), g['e_before'][i], rtol=1e-9, atol=0)
        assert np.allclose(s.last_e_after.cpu().numpy(), g['e_after'][i], rtol=1e-8, atol=0)


@pytest.mark.parametrize('n,C', [(80, 9), (40, 2051)])
def test_chi2_memo_of_the_pair_distance_log_prob(device, n, C):
    """binf_pairdist_gauss_logp_memo_f64: the plain fused log-prob bit for bit whatever
    is changed in place in between; unchanged chains are not summed again; and a
    sampler run through the class stack is identical with the memo switched off.
    2051 chains: two chains per workgroup, of which none, one or both may be unchanged."""
    from binf_amd.example import distance as DM
    ys, x = synth(n, C, 11)
    L_ = make_distance_likelihood(ys, n)
    I, J = L_.forward_model.pair_index(device)
    ty = L_.error_model.ys_device(device)
    memo = _native.new_chi2_memo(C, 3 * n, device)
    reused = lambda: memo[2][0].cpu().numpy().astype(bool)
    tx = dev_t(x, device)
    rs = np.random.RandomState(1)
    changed = np.ones(C, dtype=bool)
    for step in range(5):
        prec = 2.0 if step % 2 else dev_t(rs.uniform(1, 4, size=C), device)
        got = _native.pairdist_gauss_logp_memo(tx, I, J, ty, prec, memo)
        assert torch.equal(got, _native.pairdist_gauss_logp(tx, I, J, ty, prec)), step
        assert np.array_equal(reused(), ~changed), step
        changed = rs.rand(C) < 0.5
        idx = torch.from_numpy(np.nonzero(changed)[0]).to(device)
        tx[idx] = tx[idx] + 1e-3 * torch.randn((len(idx), 3 * n), dtype=torch.float64, device=device)
    # two entries per chain: the state (E_before) and the proposal (E_after) are both kept, so
    # the next transition's state is not summed again whichever chains were accepted
    state = tx.clone()
    for step in range(3):
        prop = state + 1e-3 * torch.randn_like(state)
        _native.pairdist_gauss_logp_memo(state, I, J, ty, 2.0, memo)
        assert reused().all() or step == 0
        _native.pairdist_gauss_logp_memo(prop, I, J, ty, 2.0, memo)
        assert not reused().any()
        acc = torch.from_numpy(rs.rand(C) < 0.5).to(device)
        state = torch.where(acc[:, None], prop, state)
    got = _native.pairdist_gauss_logp_memo(state, I, J, ty, 2.0, memo)
    assert reused().all() and torch.equal(got, _native.pairdist_gauss_logp(state, I, J, ty, 2.0))
    if C > 100:
        return
    runs = []
    for use in (True, False):
        DM.USE_CHI2_MEMO = use
        try:
            s = HMCSampler(make_post(ys, n).conditional_factory(precision=3.0), dev_t(x, device), 0.002, 4,
                           variable_name='coordinates', rng=DeviceRNG(5, device), record_energies=True)
            out = [s.sample().clone() for _ in range(4)]
            runs.append((out, s.last_e_before.clone(), s.n_accepted.clone()))
        finally:
            DM.USE_CHI2_MEMO = True
    for a, b in zip(runs[0][0], runs[1][0]):
        assert torch.equal(a, b)
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])


@pytest.mark.parametrize('n', [32, 48, 64, 65, 100, 128, 150, 192, 200, 256])
def test_packed_targets_change_no_bit(device, n):
    """binf_pairdist_pack_targets_f64 + the `_packed_` entry points: the targets in the
    order the 32..256-bead kernels hold them, read coalesced at the start of a launch
    instead of tile by tile through LDS -- force and fused leapfrog bit for bit as
    without them, for one chain per workgroup and for workgroups that walk several."""
    ys, x = synth(n, 5, 7 * n)
    lik = make_distance_likelihood(ys, n)
    em = lik.error_model
    ymat = em.ymat_device(device)
    packed = _native.pairdist_pack_targets(ymat)
    assert packed is not None and packed.numel() * 8 == _native.lib().binf_pairdist_packed_targets_bytes(n)
    assert em.ypacked_device(device) is em.ypacked_device(device)
    rs = np.random.RandomState(n)
    for C in (5, 4200 if n <= 64 else 300):
        xx = dev_t(x[:1].repeat(C, 0) + 0.1 * rs.standard_normal((C, 3 * n)), device)
        tau = dev_t(rs.uniform(0.5, 3.0, size=C), device)
        assert torch.equal(_native.pairdist_gauss_grad(xx, ymat, tau, packed=packed),
                           _native.pairdist_gauss_grad(xx, ymat, tau))
        p0 = dev_t(rs.standard_normal((C, 3 * n)), device)
        for mode in (_native.MODE_EXACT, _native.MODE_FMA):
            qa, pa, qb, pb = xx.clone(), p0.clone(), xx.clone(), p0.clone()
            _native.pairdist_leapfrog(qa, pa, ymat, tau, (0.05, 0.1), True, 2e-3, None, 5, mode)
            _native.pairdist_leapfrog(qb, pb, ymat, tau, (0.05, 0.1), True, 2e-3, None, 5, mode,
                                      packed=packed)
            assert torch.equal(qa, qb) and torch.equal(pa, pb)


@pytest.mark.parametrize('n', [257, 300, 320, 384, 448, 500, 512, 513, 576, 700, 1000, 1023, 1024])
def test_ring_kernels_every_unordered_pair_once_from_257_to_1024_beads(device, n):
    """With packed targets 257..1024 beads take the ring kernels (csrc/pairdist.hip: a wave per
    block of 64 row beads, block pairs walked in phases, every unordered pair once; odd and even
    block counts, with and without ghost beads).  The force against numpy to 1e-12 and against
    the one-sided loops (no packed targets) to 1e-13; the fused leapfrog bit for bit the
    per-step sequence over the ring gradient, in both arithmetic modes; the same bits for
    workgroups that walk several chains."""
    ys, x = synth(n, 3, 11 * n)
    lik = make_distance_likelihood(ys, n)
    ymat = lik.error_model.ymat_device(device)
    packed = _native.pairdist_pack_targets(ymat)
    assert packed is not None and packed.numel() * 8 == _native.lib().binf_pairdist_packed_targets_bytes(n)
    rs = np.random.RandomState(n)
    C = 3
    xx = dev_t(x, device)
    tau = dev_t(rs.uniform(0.5, 3.0, size=C), device)
    g = _native.pairdist_gauss_grad(xx, ymat, tau, packed=packed)
    one_sided = _native.pairdist_gauss_grad(xx, ymat, tau)
    scale = float(one_sided.abs().max())
    assert float((g - one_sided).abs().max()) <= 1e-13 * scale
    ym = ymat.cpu().numpy()
    for c in range(C):
        xc = x[c].reshape(n, 3)
        dd = xc[:, None, :] - xc[None, :, :]
        r = np.sqrt((dd ** 2).sum(-1))
        np.fill_diagonal(r, 1.0)
        w = 1.0 - ym / r
        np.fill_diagonal(w, 0.0)
        want = float(tau[c]) * (w[:, :, None] * dd).sum(1).reshape(-1)
        assert np.abs(g[c].cpu().numpy() - want).max() <= 1e-12 * np.abs(want).max()
    # many chains per workgroup: the same bits (the order of a bead's sums depends on n only)
    many = xx[:1].repeat(600, 1)
    gm = _native.pairdist_gauss_grad(many, ymat, float(tau[0]), packed=packed)
    assert torch.equal(gm, g[:1].repeat(600, 1))
    # fused leapfrog == kick / drift around the ring gradient
    p0 = dev_t(rs.standard_normal((C, 3 * n)), device)
    L, dt = 3, 2e-3
    for mode in (_native.MODE_EXACT, _native.MODE_FMA):
        qa, pa = xx.clone(), p0.clone()
        _native.pairdist_leapfrog(qa, pa, ymat, tau, (0.05, 0.1), True, dt, None, L, mode, packed=packed)
        qb, pb = xx.clone(), p0.clone()

        def force(q):
            return _native.sum_terms([_native.gauss_grad(q, 0.05, 0.1),
                                      _native.pairdist_gauss_grad(q, ymat, tau, packed=packed)])
        _native.leapfrog_kick(pb, force(qb), dt, None, half=True, mode=mode)
        _native.leapfrog_drift(qb, pb, dt, None, mode=mode)
        for _ in range(L - 1):
            _native.leapfrog_kick_drift(qb, pb, force(qb), dt, None, mode=mode)
        _native.leapfrog_kick(pb, force(qb), dt, None, half=True, mode=mode)
        assert torch.equal(qa, qb) and torch.equal(pa, pb), mode


@pytest.mark.parametrize('n', [257, 320, 384, 500, 512, 576, 1000, 1024])
def test_few_chains_take_a_wave_per_tile_same_bits_as_the_ring_kernels(device, n):
    """Few chains of 257..1024 beads (32 replicas of a 1000-bead model use an eighth of the chip with
    a workgroup per chain): with the workspace the library asks for every 64 x 64 tile of pairs is a
    wave of its own and a second launch adds a bead's partial sums in the ring kernels' order --
    force and fused leapfrog bit for bit what the ring kernels give (forced here through the C ABI
    by withholding the workspace), in both arithmetic modes, with a separate start buffer too."""
    import ctypes
    ys, x = synth(n, 5, 13 * n)
    lik = make_distance_likelihood(ys, n)
    ymat = lik.error_model.ymat_device(device)
    packed = _native.pairdist_pack_targets(ymat)
    L = _native.lib()
    C = 5
    need = L.binf_pairdist_tiles_workspace_bytes(C, n)
    assert need > 0 and L.binf_pairdist_tiles_workspace_bytes(100000, n) == 0
    assert L.binf_pairdist_tiles_workspace_bytes(C, 256) == 0 and L.binf_pairdist_tiles_workspace_bytes(C, 8193) == 0
    assert L.binf_pairdist_tiles_workspace_bytes(100000, 1025) == 0 and L.binf_pairdist_tiles_workspace_bytes(5000, 1025) > 0
    rs = np.random.RandomState(n)
    xx, tau = dev_t(x, device), dev_t(rs.uniform(0.5, 3.0, size=C), device)
    pp = lambda t: ctypes.c_void_p(t.data_ptr())
    st = _native.stream_handle(device)
    ws = torch.empty(need // 8, dtype=torch.float64, device=device)

    def grad(with_ws):
        out = torch.empty_like(xx)
        rc = L.binf_pairdist_gauss_grad_packed_f64(pp(xx), pp(ymat), pp(packed), 0.0, pp(tau), pp(out), C, n,
                                                   pp(ws) if with_ws else None, need if with_ws else 0, st)
        assert rc == 0
        return out

    assert torch.equal(grad(True), grad(False))
    assert torch.equal(_native.pairdist_gauss_grad(xx, ymat, tau, packed=packed), grad(False))   # the wrapper brings it
    p0 = dev_t(rs.standard_normal((C, 3 * n)), device)
    for mode in (_native.MODE_EXACT, _native.MODE_FMA):
        runs = []
        for with_ws in (True, False):
            for sep in (False, True):
                q = torch.empty_like(xx) if sep else xx.clone()
                p = p0.clone()
                rc = L.binf_pairdist_leapfrog_packed_f64(pp(q), pp(xx) if sep else None, pp(p), pp(ymat), pp(packed),
                                                         0.0, pp(tau), 1, 0.05, 0.1, 1, 2e-3, None, 4, C, n, mode,
                                                         pp(ws) if with_ws else None, need if with_ws else 0, st)
                assert rc == 0
                runs.append((q, p))
        for q, p in runs[1:]:
            assert torch.equal(q, runs[0][0]) and torch.equal(p, runs[0][1]), mode
    # the workspace may not overlap what the launch reads or writes
    assert L.binf_pairdist_gauss_grad_packed_f64(pp(xx), pp(ymat), pp(packed), 1.0, None, pp(ws), C, n, pp(ws), need,
                                                 st) == _native.E_ALIAS


@pytest.mark.parametrize('n,C', [(1025, 2), (1500, 3), (2048, 2), (3000, 1), (4096, 1), (8192, 1)])
def test_beyond_1024_beads_every_unordered_pair_once_as_a_wave_per_tile(device, n, C):
    """1025..8192 beads: no workgroup-per-chain form exists, the tile kernels serve alone (packed
    targets + workspace, both brought by the Python wrappers): force against the one-sided loops
    to 1e-13 and against numpy (one chain) to 1e-12; the fused leapfrog -- one tile launch and one
    update launch per force evaluation, inside ONE C-ABI call -- bit for bit the per-step sequence;
    HMCSampler takes it through the registered kind."""
    ys, x = synth(n, C, 17 * n)
    lik = make_distance_likelihood(ys, n)
    ymat = lik.error_model.ymat_device(device)
    packed = _native.pairdist_pack_targets(ymat)
    assert packed is not None and _native.lib().binf_pairdist_tiles_workspace_bytes(C, n) > 0
    rs = np.random.RandomState(n)
    xx, tau = dev_t(x, device), dev_t(rs.uniform(0.5, 3.0, size=C), device)
    g = _native.pairdist_gauss_grad(xx, ymat, tau, packed=packed)
    one_sided = _native.pairdist_gauss_grad(xx, ymat, tau)
    assert float((g - one_sided).abs().max()) <= 1e-13 * float(one_sided.abs().max())
    ym = ymat.cpu().numpy()
    xc = x[0].reshape(n, 3)
    dd = xc[:, None, :] - xc[None, :, :]
    r = np.sqrt((dd ** 2).sum(-1))
    np.fill_diagonal(r, 1.0)
    w = 1.0 - ym / r
    np.fill_diagonal(w, 0.0)
    want = float(tau[0]) * (w[:, :, None] * dd).sum(1).reshape(-1)
    assert np.abs(g[0].cpu().numpy() - want).max() <= 1e-12 * np.abs(want).max()
    p0 = dev_t(rs.standard_normal((C, 3 * n)), device)
    L, dt = 2, 1e-3
    qa, pa = xx.clone(), p0.clone()
    _native.pairdist_leapfrog(qa, pa, ymat, tau, (0.05, 0.1), True, dt, None, L, packed=packed)
    qb, pb = xx.clone(), p0.clone()
    force = lambda q: _native.sum_terms([_native.gauss_grad(q, 0.05, 0.1),
                                         _native.pairdist_gauss_grad(q, ymat, tau, packed=packed)])
    _native.leapfrog_kick(pb, force(qb), dt, None, half=True)
    _native.leapfrog_drift(qb, pb, dt, None)
    for _ in range(L - 1):
        _native.leapfrog_kick_drift(qb, pb, force(qb), dt, None)
    _native.leapfrog_kick(pb, force(qb), dt, None, half=True)
    assert torch.equal(qa, qb) and torch.equal(pa, pb)
    # without packed targets the fused call has nothing to offer beyond 1024 beads
    with pytest.raises(NotImplementedError):
        _native.pairdist_leapfrog(xx.clone(), p0.clone(), ymat, tau, None, False, dt, None, L)
    # through the class stack: the fused leapfrog of the registered kind
    prior = IsotropicGaussian(0.05, 0.1, name='coordinates_prior', variable_name='coordinates')
    cond = Posterior({lik.name: lik}, {prior.name: prior}).conditional_factory(precision=2.0)
    s = HMCSampler(cond, xx.clone(), 5e-4, 3, variable_name='coordinates', rng=DeviceRNG(1, device))
    t = HMCSampler(cond, xx.clone(), 5e-4, 3, variable_name='coordinates', rng=DeviceRNG(1, device))
    t.fused_leapfrog = False
    assert torch.equal(s.sample(), t.sample()) and bool(s.last_move_accepted.any())


@pytest.mark.parametrize('n,C', [(200, 3), (257, 5), (512, 2), (1000, 7), (1024, 1), (2048, 2), (2500, 2), (4096, 1)])
def test_chi2_by_chunks_with_few_chains_same_bits(device, n, C):
    """With fewer chains than CUs (or more than 2048 beads) the log-prob / energy entry points, given
    the workspace the library asks for, make every 8192-pair chunk of np.sum a workgroup of its own
    and add the chunk sums in order: bit for bit the workgroup-per-chain kernels (forced here through
    the C ABI by withholding the workspace), the numpy restatement on one chain, the memo variant on
    hits and misses, and the one-launch energy."""
    import ctypes
    ys, x = synth(n, C, 19 * n)
    lik = make_distance_likelihood(ys, n)
    I, J = lik.forward_model.pair_index(device)
    ty = lik.error_model.ys_device(device)
    P = I.numel()
    L = _native.lib()
    need = L.binf_pairdist_chi2_workspace_bytes(C, n, P)
    assert need > 0 and L.binf_pairdist_chi2_workspace_bytes(C, 100, 4950) == 0       # a short pair list
    assert L.binf_pairdist_chi2_workspace_bytes(100000, n, P) == 0 or n > 2048
    rs = np.random.RandomState(n)
    xx, tau = dev_t(x, device), dev_t(rs.uniform(0.5, 3.0, size=C), device)
    pp = lambda t: ctypes.c_void_p(t.data_ptr())
    st = _native.stream_handle(device)
    ws = torch.empty(need // 8, dtype=torch.float64, device=device)

    def logp(with_ws):
        out = torch.empty(C, dtype=torch.float64, device=device)
        rc = L.binf_pairdist_gauss_logp_f64(pp(xx), pp(I), pp(J), pp(ty), 0.0, pp(tau), pp(out), C, n, P,
                                            pp(ws) if with_ws else None, need if with_ws else 0, st)
        assert rc == 0
        return out

    a, b = logp(True), logp(False)
    assert torch.equal(a, b) and torch.equal(_native.pairdist_gauss_logp(xx, I, J, ty, tau), a)
    want = RD.DistancePosterior(ys, float(tau[0]), n).log_prob(coordinates=x[0]) if n <= 1024 else None
    if want is not None:
        assert float(a[0]) == want
    # the memo variant: a miss stores the chunk-summed chi^2, a hit returns it, a changed chain misses
    memo = _native.new_chi2_memo(C, 3 * n, device)
    m1 = _native.pairdist_gauss_logp_memo(xx, I, J, ty, tau, memo)
    m2 = _native.pairdist_gauss_logp_memo(xx, I, J, ty, tau * 2.0, memo)
    assert torch.equal(m1, a) and torch.equal(m2, _native.pairdist_gauss_logp(xx, I, J, ty, tau * 2.0))
    assert bool((memo[2][0] == 1).all())                                             # every chain a hit
    moved = xx.clone()
    moved[0, 0] += 1e-3
    m3 = _native.pairdist_gauss_logp_memo(moved, I, J, ty, tau, memo)
    assert torch.equal(m3, _native.pairdist_gauss_logp(moved, I, J, ty, tau)) and int(memo[2][0][0]) == 0
    # the one-launch energy (up to 2048 beads): chi^2 by chunks inside, same bits as without
    if n <= 2048:
        p0 = dev_t(rs.standard_normal((C, 3 * n)), device)
        kinds = (ctypes.c_int32 * 4)(0, 1, 1, 1)

        def energy(with_ws):
            e = torch.empty(C, dtype=torch.float64, device=device)
            lp = torch.empty(C, dtype=torch.float64, device=device)
            rc = L.binf_pairdist_hmc_energy_f64(pp(xx), pp(p0), pp(I), pp(J), pp(ty), 0.0, pp(tau), 0.05, 0.1, 2, kinds,
                                                None, 0.0, None, 0.0, pp(e), pp(lp), None, None, None, C, n, P,
                                                pp(ws) if with_ws else None, need if with_ws else 0, st)
            assert rc == 0
            return e, lp

        (e1, l1), (e2, l2) = energy(True), energy(False)
        assert torch.equal(e1, e2) and torch.equal(l1, l2)
    assert L.binf_pairdist_gauss_logp_f64(pp(xx), pp(I), pp(J), pp(ty), 1.0, None, pp(ws), C, n, P, pp(ws), need,
                                          st) == _native.E_ALIAS


def test_bead_counts_without_a_packed_form(device):
    for n in (8, 31, 8193, 9000):
        assert _native.lib().binf_pairdist_packed_targets_bytes(n) == 0
        ys, _ = synth(n, 1, n)
        em = make_distance_likelihood(ys, n).error_model
        assert em.ypacked_device(device) is None
        buf = torch.empty(16, dtype=torch.float64, device=device)
        rc = _native.lib().binf_pairdist_pack_targets_f64(em.ymat_device(device).data_ptr(), buf.data_ptr(), n,
                                                          _native.stream_handle(device))
        assert rc == _native.E_UNSUPPORTED


@pytest.mark.parametrize('n,C', [(5, 3), (17, 9), (64, 5), (100, 1030), (128, 4), (256, 7), (300, 3),
                                 (40, 2051), (131, 2048), (700, 2), (1500, 2), (2048, 1)])
def test_one_launch_energy_is_the_per_step_tier_bit_for_bit(device, n, C):
    """binf_pairdist_hmc_energy_f64 against the calls it replaces (prior row sum, chi^2,
    binf_sum_terms_f64, binf_hmc_energy_f64) for every component order, with and without
    the memo, scalar and per-chain precision; the memo it fills serves the log-prob and
    the other way round."""
    ys, x = synth(n, C, 13 * n + C)
    rs = np.random.RandomState(n + C)
    lik = make_distance_likelihood(ys, n)
    I, J = lik.forward_model.pair_index(device)
    ty = lik.error_model.ys_device(device)
    tx, tp = dev_t(x, device), dev_t(rs.standard_normal((C, 3 * n)), device)
    taus = dev_t(rs.uniform(0.5, 3.0, size=C), device)
    for prec in (2.5, taus):
        lp_lik = _native.pairdist_gauss_logp(tx, I, J, ty, prec)
        for prior, first in ((None, False), ((0.05, 0.1), True), ((0.7, -0.3), False)):
            if prior is None:
                lp = lp_lik
            else:
                lp_prior = _native.row_sum(tx, _native.ROW_SUMSQ_SHIFT, shift=prior[1], scale=-0.5 * prior[0])
                lp = _native.sum_terms([lp_prior, lp_lik] if first else [lp_lik, lp_prior])
            want = _native.hmc_energy(tp, lp)
            got, got_lp = _native.pairdist_hmc_energy(tx, tp, I, J, ty, prec, prior, first,
                                                      want_log_prob=True)
            assert torch.equal(got, want) and torch.equal(got_lp, lp)
            memo = _native.new_chi2_memo(C, 3 * n, device)
            for rep in range(2):
                assert torch.equal(_native.pairdist_hmc_energy(tx, tp, I, J, ty, prec, prior, first, memo), want)
                assert bool(memo[2][0].all()) == (rep == 1)
            assert torch.equal(_native.pairdist_gauss_logp_memo(tx, I, J, ty, prec, memo), lp_lik)
            assert bool(memo[2][0].all())
            x2 = tx.clone()
            x2[::2] += 1e-3
            _native.pairdist_gauss_logp_memo(x2, I, J, ty, prec, memo)            # second entry
            assert torch.equal(_native.pairdist_hmc_energy(tx, tp, I, J, ty, prec, prior, first, memo), want)
            assert bool(memo[2][0].all())
    # constants of the move in the Posterior's order: a value per chain and a scalar
    cvec, cs = dev_t(rs.standard_normal(C), device), -1.75
    lp_lik = _native.pairdist_gauss_logp(tx, I, J, ty, taus)
    lp_prior = _native.row_sum(tx, _native.ROW_SUMSQ_SHIFT, shift=0.1, scale=-0.5 * 0.05)
    for terms in (['lik', cvec], [cs, 'lik', cvec], ['prior', cvec, 'lik'], [cvec, 'prior', cs, 'lik'],
                  ['lik', 'prior', cs]):
        vals = [lp_lik if t == 'lik' else lp_prior if t == 'prior' else t for t in terms
                if True]
        lp = _native.sum_terms(vals)
        got, got_lp = _native.pairdist_hmc_energy(tx, tp, I, J, ty, taus, (0.05, 0.1), False, terms=terms,
                                                  want_log_prob=True)
        assert torch.equal(got_lp, lp) and torch.equal(got, _native.hmc_energy(tp, lp))


def test_sample_with_and_without_the_one_launch_energy(device):
    """HMCSampler.sample() on the restraint posterior: the fused energy changes no bit of
    states, flags, energies or adapted step sizes; a posterior with a further component
    (which the energy spec does not cover) still samples through the per-step energy."""
    n, C = 48, 11
    ys, x = synth(n, C, 21)
    runs = []
    for fused in (True, False):
        s = HMCSampler(make_post(ys, n).conditional_factory(precision=3.0), dev_t(x, device), 0.002, 4,
                       timestep_adaption_limit=3, variable_name='coordinates', rng=DeviceRNG(5, device),
                       record_energies=True)
        s.fused_energy = fused
        out = [s.sample().clone() for _ in range(5)]
        runs.append((out, s.last_e_before.clone(), s.last_e_after.clone(), s.n_accepted.clone(),
                     s.timestep.clone()))
    for a, b in zip(runs[0][0], runs[1][0]):
        assert torch.equal(a, b)
    for k in range(1, 5):
        assert torch.equal(runs[0][k], runs[1][k])
    post = make_post(ys, n)
    assert post.conditional_factory(precision=3.0).native_energy_spec('coordinates') is not None
    assert post.native_energy_spec('coordinates') is None          # precision not fixed
    # a precision prior with its variable fixed: out of the force, still a term of log_prob
    from binf_amd.example.priors import GammaPrior
    lik = make_distance_likelihood(ys, n)
    pri = IsotropicGaussian(0.05, 0.0, name='coordinates_prior', variable_name='coordinates')
    gam = GammaPrior(2.0, 0.5)
    taus = dev_t(np.random.RandomState(3).uniform(1.0, 4.0, size=C), device)
    for prec in (3.0, taus):                                   # the constant: a float / one value per chain
        cond = Posterior({lik.name: lik}, {pri.name: pri, gam.name: gam}).conditional_factory(precision=prec)
        assert cond.native_leapfrog_spec('coordinates') is not None
        spec = cond.native_energy_spec('coordinates')
        assert spec is not None and [t if isinstance(t, str) else 'const' for t in spec[-1]] == \
            ['prior', 'const', 'lik']                           # coordinates_prior < precision_prior < restraints
        runs = []
        for fused in (True, False):
            s = HMCSampler(cond, dev_t(x, device), 0.002, 4, variable_name='coordinates', rng=DeviceRNG(5, device),
                           record_energies=True)
            s.fused_energy = fused
            out = [s.sample().clone() for _ in range(3)]
            runs.append((torch.stack(out), s.last_e_before.clone(), s.last_e_after.clone(), s.n_accepted.clone()))
        assert all(torch.equal(a, b) for a, b in zip(runs[0], runs[1]))
        lp = cond.log_prob(coordinates=s.state)
        assert torch.isfinite(s.last_e_after).all() and torch.isfinite(lp).all()
    # a component with a variable still open (not a constant of the move) keeps the per-step energy
    both = Posterior({lik.name: lik}, {pri.name: pri, gam.name: gam})
    assert both.native_energy_spec('coordinates') is None


@pytest.mark.parametrize('n,C', [(48, 7), (256, 5), (300, 3), (700, 2), (20, 4)])
def test_leapfrog_reading_its_start_from_another_buffer(device, n, C):
    """binf_pairdist_leapfrog_packed_f64 with q_from: the trajectory of the in-place call on a
    copy of the start, the start left untouched -- what lets sample() skip the copy of its
    state (every force kernel: one wave per block pair, one-sided loops, with and without
    the packed targets)."""
    ys, x = synth(n, C, 5 * n)
    em = make_distance_likelihood(ys, n).error_model
    ymat, packed = em.ymat_device(device), em.ypacked_device(device)
    rs = np.random.RandomState(n)
    x0 = dev_t(x, device)
    p0 = dev_t(rs.standard_normal((C, 3 * n)), device)
    for pk in (None, packed):
        qa, pa = x0.clone(), p0.clone()
        _native.pairdist_leapfrog(qa, pa, ymat, 2.0, (0.05, 0.1), False, 2e-3, None, 4, packed=pk)
        keep = x0.clone()
        qb, pb = torch.full_like(x0, float('nan')), p0.clone()
        _native.pairdist_leapfrog(qb, pb, ymat, 2.0, (0.05, 0.1), False, 2e-3, None, 4, packed=pk, q_from=x0)
        assert torch.equal(qa, qb) and torch.equal(pa, pb) and torch.equal(x0, keep)

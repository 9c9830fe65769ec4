"""The launches bench.py times, at the sizes it times them, against the oracle.

The bench's numbers come from three call shapes that the rest of the suite only
exercises at small sizes:

* C2: ``HMCSampler.sample_n(64, thin=1, out=<one of two record buffers>)`` at 4096
  chains x 1024 dims -- the persistent kernel's no-stash path (a rejected chain reads
  its previous state back from the record buffer) with the buffers ping-ponged between
  launches;
* C5: ``HMCSampler.sample()`` on the restraint posterior at 2048 chains x 256 beads
  (two chains per workgroup in the chi^2 / energy kernel, packed targets, fused leapfrog);
* C3: ``HMCSampler.sample()`` on the polynomial coefficient conditional at 8192 chains,
  K = 33, N = 16384 (whole-tile MFMA gradient kernel inside binf_poly_leapfrog_f64).
"""
import numpy as np
import pytest
import torch

import poly_bounds as PB
from binf_amd.example.distance import make_distance_likelihood
from binf_amd.example.likelihood import POLYVAL, make_likelihood
from binf_amd.example.priors import GammaPrior, GaussianPrior
from binf_amd.pdf import IsotropicGaussian
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers.hmc import HMCSampler
from oracle import c_oracle
from oracle import ref_distance as RD
from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu


def dev_t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)


def test_c2_sample_n_with_ping_pong_record_buffers_vs_c_oracle(device):
    """bench.py:run(): sample_n(F, thin=1, p0=..., u=..., record=True, out=rec_bufs[i % 2]),
    here F = 8 and two launches, dt = 0.2 so that chains are rejected (about one in six):
    every recorded state, flag and energy equals the C oracle's transition by transition,
    and a rejected chain's row equals its previous record (its q0 for the first)."""
    C, D, L, F, dt = 4096, 1024, 20, 8, 0.2
    rs = np.random.RandomState(11)
    q0 = rs.standard_normal((C, D))
    s = HMCSampler(IsotropicGaussian(1.0, 0.0), dev_t(q0, device), dt, L, variable_name='x',
                   record_energies=True)
    rec = [torch.empty((F, C, D), dtype=torch.float64, device=device) for _ in range(2)]
    state = q0
    n_rej = 0
    for launch in range(2):
        p0 = rs.standard_normal((F, C, D))
        u = rs.uniform(size=(F, C))
        out = s.sample_n(F, thin=1, p0=dev_t(p0, device), u=dev_t(u, device), record=True,
                         out=rec[launch % 2])
        assert out.data_ptr() == rec[launch % 2].data_ptr()
        got = out.cpu().numpy()
        flags = s.accepted_history.cpu().numpy()
        eb, ea = s.last_e_before.cpu().numpy(), s.last_e_after.cpu().numpy()
        for i in range(F):
            ref = c_oracle.hmc_sample_gauss(state, p0[i], u[i], dt, L, nthreads=8)
            assert np.array_equal(flags[i].astype(np.uint8), ref['accepted']), (launch, i)
            assert np.array_equal(got[i], ref['q_out']), (launch, i)
            assert np.array_equal(eb[i], ref['e_before']) and np.array_equal(ea[i], ref['e_after'])
            rej = ~flags[i]
            n_rej += int(rej.sum())
            assert np.array_equal(got[i][rej], state[rej])         # the previous record, bit for bit
            state = ref['q_out']
        assert np.array_equal(s.state.cpu().numpy(), state)
    assert 0.02 * 16 * C < n_rej < 0.5 * 16 * C
    # the other buffer still holds the first launch's records (nothing wrote through)
    assert not torch.equal(rec[0], rec[1])


def test_c5_sample_at_2048_chains_vs_restatement(device):
    """extra.C5_2048_chains: one sample() (fused leapfrog on packed targets, one-launch energy with
    the chi^2 memo, two chains per workgroup) against RefHMCSampler on sampled chains, plus
    what must hold for all 2048: finite, accept flags consistent with the energies, the
    second call's E_before of an accepted chain = the first call's E_after (memo hit)."""
    n, C, L, dt = 256, 2048, 20, 0.002
    rs = np.random.RandomState(5)
    truth = rs.standard_normal((n, 3)) * 2.0
    ys = np.abs(RD.forward(truth.reshape(-1), n) + 0.05 * rs.standard_normal(n * (n - 1) // 2))
    x = truth.reshape(-1)[None, :] + 0.1 * rs.standard_normal((C, 3 * n))
    p0 = rs.standard_normal((C, 3 * n))
    u = rs.uniform(size=C)
    lik = make_distance_likelihood(ys, n)
    prior = IsotropicGaussian(0.05, 0.0, name='coordinates_prior', variable_name='coordinates')
    cond = Posterior({lik.name: lik}, {prior.name: prior}).conditional_factory(precision=4.0)
    assert cond.native_leapfrog_spec('coordinates') is not None
    assert cond.native_energy_spec('coordinates') is not None
    s = HMCSampler(cond, dev_t(x, device), dt, L, variable_name='coordinates')
    out = s.sample(p0=dev_t(p0, device), u=dev_t(u, device)).cpu().numpy()
    acc = s.last_move_accepted.cpu().numpy()
    eb, ea = s.last_e_before.cpu().numpy(), s.last_e_after.cpu().numpy()
    assert np.isfinite(out).all() and np.isfinite(eb).all() and np.isfinite(ea).all()
    assert np.array_equal(acc, u < np.exp(np.clip(-(ea - eb), -308.0, 709.0)))
    assert 0.9 < acc.mean() <= 1.0
    for c in (0, 1, 777, 1024, 2046, 2047):                 # both chains of a workgroup, both ends
        ref = R.RefHMCSampler(RD.DistancePosterior(ys, 4.0, n, prior_k=0.05), x[c].copy(), dt, L,
                              variable_name='coordinates', normal=lambda size, c=c: p0[c].copy(),
                              uniform=lambda c=c: u[c])
        want = ref.sample()
        assert bool(acc[c]) == bool(ref.last_move_accepted)
        assert np.abs(out[c] - want).max() <= 1e-9 * np.abs(want).max()
        assert abs(eb[c] - ref.last_E_before) <= 1e-10 * abs(ref.last_E_before)
        assert abs(ea[c] - ref.last_E_after) <= 1e-9 * abs(ref.last_E_after)
    # a second transition: its E_before is -log_prob of the state the first one ended with (the
    # chi^2 memo's hit for accepted chains, its other entry for rejected ones) + the new kinetic
    # energy, in numpy's order: the same bits as the stand-alone log-prob kernel + np.sum
    p1 = rs.standard_normal((C, 3 * n))
    s.sample(p0=dev_t(p1, device), u=dev_t(rs.uniform(size=C), device))
    eb2 = s.last_e_before.cpu().numpy()
    lp = cond.log_prob(coordinates=dev_t(out, device)).cpu().numpy()
    want = np.array([-lp[c] + 0.5 * np.sum(p1[c] ** 2) for c in range(C)])
    assert np.array_equal(eb2, want)


def test_c3_sample_at_8192_chains_inside_the_bound(device):
    """extra.C3: one HMCSampler.sample() at 8192 chains, K = 33, N = 16384, L = 20 (21 launches of
    the whole-tile MFMA gradient kernel + partial-sum / kick / drift, one Horner pass per energy)
    against RefHMCSampler on sampled chains inside the propagated 1e-10 bound; finite and accepted
    everywhere (dt = 2e-4 is far inside the stable step)."""
    K, N, C, L, dt, tau = 33, 16384, 8192, 20, 2e-4, 2.5
    xs = np.linspace(-1, 1, N)
    ys = POLYVAL(xs, np.random.RandomState(7).standard_normal(K)) + \
        np.random.RandomState(9).standard_normal(N) / np.sqrt(tau)
    rs = np.random.RandomState(8)
    q0 = rs.standard_normal((C, K))
    p0 = rs.standard_normal((C, K))
    u = rs.uniform(size=C)
    lik = make_likelihood(xs, ys, POLYVAL)
    post = Posterior({lik.name: lik},
                     {'precision_prior': GammaPrior(1.0, 0.2),
                      'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K) * 5)})
    cond = post.conditional_factory(precision=tau)
    s = HMCSampler(cond, dev_t(q0, device), dt, L, variable_name='coefficients', record_energies=True)
    assert s._fused_spec('coefficients', K, C) is None          # the per-step tier with the MFMA gradient
    out = s.sample(p0=dev_t(p0, device), u=dev_t(u, device)).cpu().numpy()
    acc = s.last_move_accepted.cpu().numpy()
    eb, ea = s.last_e_before.cpu().numpy(), s.last_e_after.cpu().numpy()
    assert np.isfinite(out).all() and np.isfinite(ea).all()
    assert np.array_equal(acc, u < np.exp(np.clip(-(ea - eb), -308.0, 709.0)))
    pb = PB.PolyBound(xs, ys, K, prior_mu=np.zeros(K), prior_var=np.ones(K) * 5)
    for c in (0, 63, 64, 4095, 4096, 8191):                    # both tiles of a wave, both ends
        ref = R.RefHMCSampler(R.PolyCoefficientsConditional(xs, ys, tau, np.zeros(K), np.ones(K) * 5, 1.0,
                                                            1.0),   # rate 1.0: quirk Q6 (clone passes shape twice)
                              q0[c].copy(), dt, L, variable_name='coefficients',
                              normal=lambda size, c=c: p0[c].copy(), uniform=lambda c=c: u[c])
        want = ref.sample()
        b = pb.transition(q0[c], p0[c], tau, dt, L)
        if abs(u[c] - np.exp(-(ref.last_E_after - ref.last_E_before))) > 1e-6:
            assert bool(acc[c]) == bool(ref.last_move_accepted)
        if acc[c] and ref.last_move_accepted:
            assert np.all(np.abs(out[c] - want) <= b['bq']), c
        assert abs(eb[c] - ref.last_E_before) <= b['be_before'] + 1e-10 * abs(ref.last_E_before)
        assert abs(ea[c] - ref.last_E_after) <= b['be_after'] + 1e-10 * abs(ref.last_E_after)

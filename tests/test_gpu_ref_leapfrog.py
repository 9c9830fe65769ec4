"""The HIP integrators against outputs of THE REFERENCE'S OWN ``_leapfrog``
(``binf/samplers/hmc.py:92-125``; fixtures ``tests/golden/ref_leapfrog_*.npz``, made
by ``oracle/gen_ref_leapfrog.py`` from the reference's source with its csb import
dropped -- see ``tests/test_ref_leapfrog.py``).

* Gaussian (``k * (x - x0)``, every operation numpy's): ``HMCSampler._leapfrog`` --
  the per-step tier -- returns the reference's q AND p bit for bit; the fused
  trajectory kernels (persistent / split / long-chain, through ``sample()`` with
  u = 0 so that every proposal is accepted) return the reference's q bit for bit and
  an E_after that equals numpy's V(q) + 0.5 sum(p**2) of the reference's end state.
* polynomial coefficient conditional: the force contraction is BLAS in the reference,
  so the end state is held inside the propagated 1e-10 bound of ``poly_bounds``
  (fused polynomial leapfrog and per-step tier).
* pair-distance posterior (build-defined PDF, reference integrator): 1e-10.
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
from conftest import golden_files, load_golden

pytestmark = pytest.mark.gpu

FILES = golden_files('ref_leapfrog_')
ident = lambda p: p.split('ref_leapfrog_')[-1][:-4]


def dev_t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)


def _timestep(g, device):
    t = np.asarray(g['timestep'], dtype=np.float64)
    return float(t) if t.ndim == 0 else dev_t(t, device)


@pytest.mark.parametrize('path', [f for f in FILES if 'gauss' in f], ids=ident)
def test_per_step_tier_returns_the_reference_q_and_p_bitwise(device, path):
    g = load_golden(path)
    pdf = IsotropicGaussian(float(g['k']), float(g['x0']))
    s = HMCSampler(pdf, dev_t(g['q0'], device), 0.1, int(g['nsteps']), variable_name='x')
    q, p = dev_t(g['q0'], device), dev_t(g['p0'], device)
    rq, rp = s._leapfrog(q, p, _timestep(g, device), int(g['nsteps']))
    assert rq is q and rp is p                                    # in place, hmc.py:116-125
    assert np.array_equal(q.cpu().numpy(), g['q_out'])
    assert np.array_equal(p.cpu().numpy(), g['p_out'])
    # one chain as a [D] vector, the reference's own call shape
    q1, p1 = dev_t(g['q0'][0], device), dev_t(g['p0'][0], device)
    t = np.asarray(g['timestep']).reshape(-1)[0]
    s._leapfrog(q1, p1, float(t), int(g['nsteps']))
    assert np.array_equal(q1.cpu().numpy(), g['q_out'][0]) and np.array_equal(p1.cpu().numpy(), g['p_out'][0])


@pytest.mark.parametrize('path', [f for f in FILES if 'gauss' in f], ids=ident)
def test_fused_trajectory_kernels_reproduce_the_reference_integrator(device, path):
    g = load_golden(path)
    k, x0, L = float(g['k']), float(g['x0']), int(g['nsteps'])
    C, D = g['q0'].shape
    want_e = np.array([0.5 * k * np.sum((g['q_out'][c] - x0) ** 2) + 0.5 * np.sum(g['p_out'][c] ** 2)
                       for c in range(C)])
    s = HMCSampler(IsotropicGaussian(k, x0), dev_t(g['q0'], device), 0.1, L, variable_name='x',
                   record_energies=True)
    s.timestep = _timestep(g, device)
    out = s.sample(p0=dev_t(g['p0'], device), u=torch.zeros(C, dtype=torch.float64, device=device))
    assert bool(s.last_move_accepted.all())
    assert np.array_equal(out.cpu().numpy(), g['q_out'])
    assert np.array_equal(s.last_e_after.cpu().numpy(), want_e)
    # the multi-transition launch, n = 1
    s2 = HMCSampler(IsotropicGaussian(k, x0), dev_t(g['q0'], device), 0.1, L, variable_name='x',
                    record_energies=True)
    s2.timestep = _timestep(g, device)
    rec = s2.sample_n(1, p0=dev_t(g['p0'], device)[None], u=torch.zeros((1, C), dtype=torch.float64,
                                                                       device=device))
    assert np.array_equal(rec[0].cpu().numpy(), g['q_out'])
    assert np.array_equal(s2.last_e_after.reshape(-1).cpu().numpy(), want_e)


@pytest.mark.parametrize('path', [f for f in FILES if 'poly' in f], ids=ident)
@pytest.mark.parametrize('fused', [True, False])
def test_polynomial_leapfrog_inside_the_propagated_bound(device, path, fused):
    g = load_golden(path)
    C, K = g['q0'].shape
    xs, ys, tau, dt, L = g['xs'], g['ys'], float(g['precision']), float(g['timestep']), int(g['nsteps'])
    lik = make_likelihood(xs, ys, POLYVAL)
    post = Posterior({lik.name: lik}, {'precision_prior': GammaPrior(1.0, 0.2),
                                       'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K))})
    cond = post.conditional_factory(precision=tau)
    s = HMCSampler(cond, dev_t(g['q0'], device), dt, L, variable_name='coefficients')
    s.fused_leapfrog = fused
    q, p = dev_t(g['q0'], device), dev_t(g['p0'], device)
    s._leapfrog(q, p, dt, L)
    q, p = q.cpu().numpy(), p.cpu().numpy()
    pb = PB.PolyBound(xs, ys, K)
    for c in range(C):
        b = pb.transition(g['q0'][c], g['p0'][c], tau, dt, L)
        assert np.all(np.abs(q[c] - g['q_out'][c]) <= b['bq']), (c, np.abs(q[c] - g['q_out'][c]) / b['bq'])
        assert np.all(np.abs(p[c] - g['p_out'][c]) <= b['bp']), (c, np.abs(p[c] - g['p_out'][c]) / b['bp'])
    assert not np.array_equal(q, g['q0'])


@pytest.mark.parametrize('path', [f for f in FILES if 'dist' in f], ids=ident)
@pytest.mark.parametrize('fused', [True, False])
def test_pair_distance_leapfrog_vs_the_reference_integrator(device, path, fused):
    g = load_golden(path)
    n, tau, dt, L = int(g['n_beads']), float(g['precision']), float(g['timestep']), int(g['nsteps'])
    lik = make_distance_likelihood(g['ys'], n)
    priors = {}
    if float(g['prior_k']):
        priors['coordinates_prior'] = IsotropicGaussian(float(g['prior_k']), 0.0, name='coordinates_prior',
                                                        variable_name='coordinates')
    cond = Posterior({lik.name: lik}, priors).conditional_factory(precision=tau)
    s = HMCSampler(cond, dev_t(g['q0'], device), dt, L, variable_name='coordinates')
    s.fused_leapfrog = fused
    assert (cond.native_leapfrog_spec('coordinates') is not None)
    q, p = dev_t(g['q0'], device), dev_t(g['p0'], device)
    s._leapfrog(q, p, dt, L)
    q, p = q.cpu().numpy(), p.cpu().numpy()
    assert np.abs(q - g['q_out']).max() <= 1e-10 * np.abs(g['q_out']).max()
    assert np.abs(p - g['p_out']).max() <= 1e-10 * np.abs(g['p_out']).max()
    assert not np.array_equal(q, g['q0'])


@pytest.mark.parametrize('path', golden_files('ref_adapt_timestep_'),
                         ids=lambda p: p.split('ref_adapt_timestep_')[-1][:-4])
@pytest.mark.parametrize('D', [8, 1024, 9000])
def test_adaption_follows_the_reference_run_bitwise(device, path, D):
    """Step-size adaption inside the kernels (persistent kernel D = 8 / 1024, long-chain path
    D = 9000) and the ``_adapt_timestep`` method against the step sizes the REFERENCE's own
    ``_adapt_timestep`` (hmc.py:183-191) produced for the same accept / reject sequences; the
    moves are forced with u = 0 (accept) / u = inf (reject)."""
    g = load_golden(path)
    flags, want = g['accepted'], g['timesteps']
    up, down, dt0 = float(g['uprate']), float(g['downrate']), float(g['timestep0'])
    C, n = flags.shape
    u = dev_t(np.where(flags.T, 0.0, np.inf), device)                       # [n, C]
    q0 = torch.zeros((C, D), dtype=torch.float64, device=device)
    kw = dict(timestep_adaption_limit=1000, adaption_uprate=up, adaption_downrate=down, variable_name='x')
    # n single calls
    s = HMCSampler(IsotropicGaussian(), q0, dt0, 1, **kw)
    for i in range(n):
        s.sample(p0=torch.full((C, D), 1e-3, dtype=torch.float64, device=device), u=u[i])
        assert np.array_equal(s.last_move_accepted.cpu().numpy(), flags[:, i])
        assert np.array_equal(s.timestep.cpu().numpy(), want[:, i]), i
    # one multi-transition launch
    s = HMCSampler(IsotropicGaussian(), q0, dt0, 1, **kw)
    s.sample_n(n, p0=torch.full((n, C, D), 1e-3, dtype=torch.float64, device=device), u=u, record=False)
    assert np.array_equal(s.timestep.cpu().numpy(), want[:, -1])
    # the method, driven by hand as the fixture was made
    s = HMCSampler(IsotropicGaussian(), q0, dt0, 1, **kw)
    for i in range(n):
        s._last_move_accepted = torch.from_numpy(flags[:, i]).to(device)
        s._adapt_timestep()
        assert np.array_equal(s.timestep.cpu().numpy(), want[:, i]), i


@pytest.mark.parametrize('path', golden_files('ref_sample_'), ids=lambda p: p.split('ref_sample_')[-1][:-4])
def test_whole_transitions_vs_the_reference_run_sample(device, path):
    """``tests/golden/ref_sample_*.npz``: the reference's own ``sample()`` run statement by statement
    (all but the csb line :151; see tests/test_ref_leapfrog.py) with the draws it made recorded.  The
    fused kernels -- persistent kernel (one call per transition, and all transitions in one launch),
    long-chain path for D = 9000 -- fed with those draws return the reference's energies, accept
    flags, states, adapted step sizes and counters bit for bit."""
    g = load_golden(path)
    C, ncalls, D = g['p0'].shape
    k, x0, L = float(g['k']), float(g['x0']), int(g['nsteps'])
    limit, dt0 = int(g['adaption_limit']), float(g['timestep0'])
    p0 = dev_t(np.swapaxes(g['p0'], 0, 1), device)                  # [ncalls, C, D]
    u = dev_t(g['u'].T, device)                                     # [ncalls, C]

    def sampler():
        return HMCSampler(IsotropicGaussian(k, x0), dev_t(g['q0'], device), dt0, L,
                          timestep_adaption_limit=limit, variable_name='x', record_energies=True)
    s = sampler()
    for i in range(ncalls):
        out = s.sample(p0=p0[i], u=u[i])
        assert np.array_equal(s.last_e_before.cpu().numpy(), g['e_before'][:, i]), i
        assert np.array_equal(s.last_e_after.cpu().numpy(), g['e_after'][:, i]), i
        assert np.array_equal(s.last_move_accepted.cpu().numpy(), g['accepted'][:, i]), i
        assert np.array_equal(out.cpu().numpy(), g['state'][:, i]), i
        ts = s.timestep
        ts = ts.cpu().numpy() if isinstance(ts, torch.Tensor) else np.full(C, ts)
        assert np.array_equal(ts, g['timestep'][:, i]), i
        assert s.counter == int(g['counter'][0, i])
        assert np.array_equal(s.n_accepted.cpu().numpy(), g['n_accepted'][:, i])
    s = sampler()
    rec = s.sample_n(ncalls, p0=p0, u=u)
    assert np.array_equal(rec.cpu().numpy(), np.swapaxes(g['state'], 0, 1))
    assert np.array_equal(s.accepted_history.cpu().numpy(), g['accepted'].T)
    assert np.array_equal(s.last_e_after.cpu().numpy(), g['e_after'].T)
    ts = s.timestep
    ts = ts.cpu().numpy() if isinstance(ts, torch.Tensor) else np.full(C, ts)
    assert np.array_equal(ts, g['timestep'][:, -1])
    # ... and the per-step tier (any pdf.gradient / pdf.log_prob): the same bits
    s = sampler()
    s.pdf.native_hmc_spec = lambda name: None
    for i in range(ncalls):
        out = s.sample(p0=p0[i], u=u[i])
        assert np.array_equal(out.cpu().numpy(), g['state'][:, i]), i
        assert np.array_equal(s.last_e_after.cpu().numpy(), g['e_after'][:, i]), i

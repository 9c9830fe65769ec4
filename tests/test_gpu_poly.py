"""GPU parity tests of the polynomial forward model + Gaussian error model
path (BASELINE configs C1 / C3 / C4) through the C ABI and through the
Posterior / Likelihood / Gibbs class stack.

Bars: Horner forward and chi^2 are BIT-EXACT against numpy; everything that
involves log() of the precision or the J.r contraction (numpy: BLAS dgemv,
whose summation order is not reproducible) is held to 1e-10 relative
(BASELINE.json north_star), accept flags identical.  For whole trajectories
"1e-10" is made precise by tests/poly_bounds.py: every force evaluation may
differ from numpy's by 1e-10 of its sum-of-magnitudes scale, and that error is
propagated through the (linear) leapfrog map exactly -- the asserted tolerance
of every state and energy below is that computed bound, not a flat rtol."""
import numpy as np
import pytest
import torch

from binf_amd import _native
from binf_amd.example.likelihood import (POLYVAL, ForwardModel,
                                         GaussianErrorModel, make_likelihood)
from binf_amd.example.misc import make_posterior
from binf_amd.example.priors import GammaPrior, GaussianPrior
from binf_amd.example.samplers import (GammaSampler, RWMCSampler,
                                       make_hmc_sampler, make_sampler)
from binf_amd.samplers import BinfState
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.pdf.posteriors import Posterior
import poly_bounds as PB
from conftest import golden_files, load_golden
from oracle import ref_example as RE
from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def dev_t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)


def rel_close(got, want, rtol=RTOL):
    got, want = np.asarray(got), np.asarray(want)
    scale = np.abs(want).max() if want.size else 1.0
    return np.abs(got - want).max() <= rtol * max(scale, 1e-300)


def synth(K, N, C, seed, xlim=1.0, tau=2.5):
    rs = np.random.RandomState(seed)
    xs = np.linspace(-xlim, xlim, N)
    c_true = rs.standard_normal(K)
    ys = R.polyval(xs, c_true) + rs.standard_normal(N) / np.sqrt(tau)
    theta = c_true + 0.3 * rs.standard_normal((C, K))
    return xs, ys, theta


SHAPES = [(1, 1, 3), (17, 50, 5), (18, 33, 70), (34, 300, 9), (49, 200, 5), (50, 64, 3), (4, 20, 5), (4, 20, 70), (7, 37, 9), (8, 16, 64),
          (17, 100, 33), (33, 1000, 20), (33, 16384, 6), (36, 129, 3),
          (48, 500, 4), (64, 8200, 2), (5, 7700, 3), (3, 15892, 2)]


@pytest.mark.parametrize('K,N,C', SHAPES)
def test_poly_forward_is_numpy_polyval_bitwise(device, K, N, C):
    xs, ys, theta = synth(K, N, C, K * 100 + N, xlim=2.0 if K <= 8 else 1.0)
    got = _native.poly_forward(dev_t(theta, device), dev_t(xs, device)).cpu().numpy()
    want = np.stack([POLYVAL(xs, theta[c]) for c in range(C)])
    assert np.array_equal(got, want)


@pytest.mark.parametrize('K,N,C', SHAPES)
def test_error_model_and_fused_logp(device, K, N, C):
    xs, ys, theta = synth(K, N, C, K * 7 + N)
    mock = np.stack([POLYVAL(xs, theta[c]) for c in range(C)])
    tmock, tys = dev_t(mock, device), dev_t(ys, device)
    # chi^2 bit-exact: at precision 1 the log-prob is exactly -0.5*chi2
    want1 = np.array([-0.5 * np.sum((mock[c] - ys) ** 2) * 1.0 +
                      N * 0.5 * np.log(1.0) for c in range(C)])
    assert np.array_equal(_native.gauss_err_logp(tmock, tys, 1.0).cpu().numpy(), want1)
    assert np.array_equal(
        _native.poly_gauss_logp(dev_t(theta, device), dev_t(xs, device), tys, 1.0)
        .cpu().numpy(), want1)
    # general precision, scalar and per chain (device log): tolerance
    taus = np.random.RandomState(1).uniform(0.5, 4.0, size=C)
    for prec, tarr in ((2.5, np.full(C, 2.5)), (dev_t(taus, device), taus)):
        want = np.array([-0.5 * np.sum((mock[c] - ys) ** 2) * tarr[c] +
                         N * 0.5 * np.log(tarr[c]) for c in range(C)])
        assert rel_close(_native.gauss_err_logp(tmock, tys, prec).cpu().numpy(), want, 1e-13)
        assert rel_close(_native.poly_gauss_logp(dev_t(theta, device), dev_t(xs, device),
                                                 tys, prec).cpu().numpy(), want, 1e-13)
    # error-model gradient: elementwise, exact
    got = _native.gauss_err_grad(tmock, tys, dev_t(taus, device)).cpu().numpy()
    assert np.array_equal(got, (mock - ys) * taus[:, None])


# data sets made of whole 16-point tiles take the trimmed kernel (poly_grad_mfma_full_kernel): one
# tile, fewer tiles than splits, every K instantiation incl. the VALU tails (17, 18, 33, 34, 49,
# 50), ragged chain counts on either side of the two-tiles-per-wave threshold (4096), C = 1
WHOLE_TILE_SHAPES = [(1, 16, 1), (4, 32, 5), (8, 16, 200), (16, 48, 3), (17, 64, 130), (18, 16, 70),
                     (32, 160, 9), (33, 64, 130), (34, 16, 200), (36, 320, 7), (48, 96, 65), (49, 32, 3),
                     (50, 160, 70), (64, 320, 9), (16, 4096, 4100), (33, 1024, 4097), (5, 16384, 2)]


@pytest.mark.parametrize('K,N,C', SHAPES + [(33, 4096, 130), (4, 20, 1000)] + WHOLE_TILE_SHAPES)
def test_mfma_gradient_matches_numpy_chain_rule(device, K, N, C):
    """J . ((mock - ys) * tau) with J = vstack([xs**i]) (likelihoods.py:148-155)."""
    xs, ys, theta = synth(K, N, C, K + N + C)
    fwm = ForwardModel(xs, POLYVAL)
    A = fwm.design_matrix(K, device)
    assert np.array_equal(A.cpu().numpy(), np.vstack([xs ** i for i in range(K)]))
    taus = np.random.RandomState(2).uniform(0.5, 4.0, size=C)
    Jn = np.vstack([xs ** i for i in range(K)])
    for prec, tarr in ((2.5, np.full(C, 2.5)), (dev_t(taus, device), taus)):
        got = _native.poly_gauss_grad(dev_t(theta, device), A, dev_t(ys, device),
                                      prec).cpu().numpy()
        want = np.stack([Jn.dot((POLYVAL(xs, theta[c]) - ys) * tarr[c])
                         for c in range(C)])
        # per-row scale: sum |J| |r| bounds the rounding of the contraction
        bound = np.stack([np.abs(Jn).dot(np.abs((POLYVAL(xs, theta[c]) - ys) * tarr[c]))
                          for c in range(C)])
        assert np.all(np.abs(got - want) <= RTOL * np.maximum(bound, 1e-300))
    # deterministic: two runs give identical bits
    g1 = _native.poly_gauss_grad(dev_t(theta, device), A, dev_t(ys, device), 2.5)
    g2 = _native.poly_gauss_grad(dev_t(theta, device), A, dev_t(ys, device), 2.5)
    assert torch.equal(g1, g2)


def test_likelihood_stack_dispatches_to_fused_kernels(device):
    K, N, C = 4, 20, 12
    xs, ys, theta = synth(K, N, C, 5, xlim=2.0)
    L = make_likelihood(xs, ys, POLYVAL)
    assert L.variables == {'coefficients', 'precision'}
    tc = dev_t(theta, device)
    lp = L.log_prob(coefficients=tc, precision=2.5).cpu().numpy()
    gr = L.gradient(coefficients=tc, precision=2.5).cpu().numpy()
    Jn = np.vstack([xs ** i for i in range(K)])
    for c in range(C):
        mock = POLYVAL(xs, theta[c])
        assert abs(lp[c] - (-0.5 * np.sum((mock - ys) ** 2) * 2.5 +
                            N * 0.5 * np.log(2.5))) <= 1e-12 * abs(lp[c])
        assert np.allclose(gr[c], Jn.dot((mock - ys) * 2.5), rtol=1e-10, atol=1e-12)
    # the unfused plug-in route (models called as written) agrees
    mock_t = L.forward_model(coefficients=tc)
    lp2 = L.error_model.log_prob(mock_data=mock_t, precision=2.5)
    assert np.allclose(lp2.cpu().numpy(), lp, rtol=1e-13)
    from binf_amd.pdf.likelihoods import contract_jacobian
    g2 = contract_jacobian(L.forward_model.jacobi_matrix(coefficients=tc),
                           L.error_model.gradient(mock_data=mock_t, precision=2.5))
    assert np.allclose(g2.cpu().numpy(), gr, rtol=1e-10, atol=1e-12)


def test_non_numpy_polynomial_callable_is_applied_as_given(device):
    xs = np.linspace(-1, 1, 10)
    fwm = ForwardModel(dev_t(xs, device), lambda x, c: c[..., :1] + 2.0 * x)
    assert fwm.native_spec() is None
    c = torch.ones((3, 2), dtype=torch.float64, device=device)
    assert torch.allclose(fwm(coefficients=c)[0], 1.0 + 2.0 * dev_t(xs, device))


def test_priors_match_reference_formulas_and_quirks(device):
    K, C = 4, 9
    rs = np.random.RandomState(0)
    theta = rs.standard_normal((C, K))
    means, var = rs.standard_normal(K), rs.uniform(1, 5, K)
    cp = GaussianPrior(means, var)
    got = cp.log_prob(coefficients=dev_t(theta, device)).cpu().numpy()
    want = np.array([-0.5 * np.sum((theta[c] - means) ** 2 / var) for c in range(C)])
    assert np.array_equal(got, want)
    assert cp.differentiable_variables == set()            # quirk Q4
    gp = GammaPrior(1.0, 0.2)
    tau = dev_t(rs.uniform(0.5, 3, C), device)
    assert torch.allclose(gp.log_prob(precision=tau), (1.0 - 1.0) * torch.log(tau) - tau * 0.2)
    cl = gp.clone()
    assert cl.shape == 1.0 and cl.rate == 1.0             # quirk Q6


def test_conditional_posterior_matches_restatement(device):
    """Energy = all three components in sorted-name order; force = likelihood
    only (quirk Q4)."""
    xs, ys = RE.example_data()
    K, C = 4, 16
    theta = np.random.RandomState(3).standard_normal((C, K))
    post = make_posterior(xs, ys, POLYVAL)
    assert post.variables == {'coefficients', 'precision'}
    cond = post.conditional_factory(precision=2.0)
    assert cond.variables == {'coefficients'}
    tc = dev_t(theta, device)
    lp = cond.log_prob(coefficients=tc).cpu().numpy()
    gr = cond.gradient(coefficients=tc).cpu().numpy()
    for c in range(C):
        ref = RE.conditional_pdf(xs, ys, 2.0, K)
        assert abs(lp[c] - ref.log_prob(coefficients=theta[c])) <= 1e-12 * abs(lp[c])
        assert np.allclose(gr[c], ref.gradient(coefficients=theta[c]), rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize('K,N,C,L,dt,xlim', [(4, 20, 24, 50, 0.02, 2.0),
                                             (33, 2048, 6, 5, 2e-4, 1.0)])
def test_hmc_on_polynomial_posterior_vs_restatement(device, K, N, C, L, dt, xlim):
    """Generic tier around the fused log-prob / MFMA gradient kernels against
    the numpy restatement of HMCSampler on the same conditional posterior."""
    if K == 4:
        xs, ys = RE.example_data(N)
    else:
        xs, ys, _ = synth(K, N, 1, 11, xlim)
    rs = np.random.RandomState(K)
    q0 = 0.1 * rs.standard_normal((C, K)) + (1.0 if K == 4 else 0.0)
    p0 = rs.standard_normal((C, K))
    u = rs.uniform(size=C)
    post = make_posterior(xs, ys, POLYVAL) if K == 4 else None
    if post is None:
        from binf_amd.pdf.posteriors import Posterior
        lik = make_likelihood(xs, ys, POLYVAL)
        post = Posterior({lik.name: lik},
                         {'precision_prior': GammaPrior(1.0, 0.2),
                          'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K) * 5)})
    cond = post.conditional_factory(precision=2.5)
    s = HMCSampler(cond, dev_t(q0, device), dt, L, variable_name='coefficients')
    s.fused_transition = False                    # this test is the per-step tier
    out = s.sample(p0=dev_t(p0, device), u=dev_t(u, device)).cpu().numpy()
    acc = s.last_move_accepted.cpu().numpy()
    eb, ea = s.last_e_before.cpu().numpy(), s.last_e_after.cpu().numpy()
    pb = PB.PolyBound(xs, ys, K, np.zeros(K), np.ones(K) * 5)
    margin = 0.0
    for c in range(C):
        ref = R.RefHMCSampler(RE.conditional_pdf(xs, ys, 2.5, K), q0[c].copy(), dt, L,
                              variable_name='coefficients',
                              normal=lambda size, c=c: p0[c].copy(),
                              uniform=lambda c=c: u[c])
        want = ref.sample()
        b = pb.transition(q0[c], p0[c], 2.5, dt, L)
        assert bool(acc[c]) == bool(ref.last_move_accepted), c
        if acc[c]:
            assert np.all(np.abs(out[c] - want) <= b['bq'] + 4 * PB.U * np.abs(want))
            margin = max(margin, np.max(np.abs(out[c] - want) / b['bq']))
        else:
            assert np.array_equal(out[c], want)
        assert abs(eb[c] - ref.last_E_before) <= b['be_before']
        assert abs(ea[c] - ref.last_E_after) <= b['be_after']
    assert 0 < acc.mean()
    # how much of the 1e-10 allowance the kernel actually uses (typically 1e-5 of it)
    assert margin < 1.0


def _small_posterior(xs, ys, K, prior, lik_name='points'):
    from binf_amd.pdf.likelihoods import Likelihood
    from binf_amd.pdf.posteriors import Posterior
    lik = Likelihood(lik_name, ForwardModel(xs, POLYVAL), GaussianErrorModel(ys))
    priors = {'precision_prior': GammaPrior(1.0, 0.2)}
    if prior:
        priors['coefficients_prior'] = GaussianPrior(np.linspace(-0.5, 0.5, K),
                                                     np.linspace(2.0, 5.0, K))
    return Posterior({lik.name: lik}, priors)


@pytest.mark.parametrize('K,N,C,L,prior,lik_name,mode', [
    (4, 20, 70, 50, True, 'points', 'exact'),      # example_script.py's shape
    (4, 20, 5, 7, True, 'a_first', 'exact'),       # likelihood sorts before the prior
    (1, 1, 3, 3, False, 'points', 'exact'),
    (3, 7, 65, 4, True, 'points', 'fma'),
    (7, 8, 9, 4, True, 'points', 'exact'),
    (8, 37, 130, 5, True, 'points', 'exact'),
    (9, 100, 6, 3, False, 'points', 'exact'),
    (13, 127, 4, 3, True, 'points', 'fma'),
    (16, 128, 66, 2, True, 'points', 'exact'),
    # more than 128 data points: one wave per chain (csrc/hmc_poly_wave.hip)
    (4, 129, 9, 5, True, 'points', 'exact'),
    (8, 258, 70, 3, True, 'a_first', 'exact'),
    (16, 700, 5, 2, False, 'points', 'fma'),
    (3, 1024, 33, 4, True, 'points', 'exact')])
def test_fused_small_polynomial_transition_vs_per_step_tier(device, K, N, C, L, prior,
                                                            lik_name, mode):
    """binf_hmc_sample_poly_f64 (one launch) against the per-step tier on the
    same conditional posterior and draws: E_before identical to the bit (same
    state, numpy's summation order in both), trajectory / E_after to 1e-10
    (the force is an FMA dot product here, MFMA there), same accept flags;
    per-chain precision and per-chain adapting step sizes included."""
    rs = np.random.RandomState(100 * K + N)
    xs = np.linspace(-1.5, 1.5, N)
    ys = R.polyval(xs, rs.standard_normal(K)) + 0.5 * rs.standard_normal(N)
    q0 = 0.2 * rs.standard_normal((C, K))
    p0 = rs.standard_normal((2, C, K))
    u = rs.uniform(size=(2, C))
    tau = 0.5 + rs.uniform(size=C)
    dt = 0.02 / max(1.0, N / 20.0) / K
    post = _small_posterior(xs, ys, K, prior, lik_name)
    cond = post.conditional_factory(precision=dev_t(tau, device))
    assert cond.native_hmc_spec('coefficients') is not None
    got = {}
    for fused in (True, False):
        s = HMCSampler(cond, dev_t(q0, device), dt, L, timestep_adaption_limit=10,
                       variable_name='coefficients', mode=mode)
        s.fused_transition = fused
        res = []
        for i in range(2):
            out = s.sample(p0=dev_t(p0[i], device), u=dev_t(u[i], device))
            res.append((out.cpu().numpy().copy(), s.last_move_accepted.cpu().numpy().copy(),
                        s.last_e_before.cpu().numpy().copy(),
                        s.last_e_after.cpu().numpy().copy(), s.timestep.cpu().numpy().copy()))
        got[fused] = (res, s.n_accepted.cpu().numpy())
    pb = PB.PolyBound(xs, ys, K, *((np.linspace(-0.5, 0.5, K), np.linspace(2.0, 5.0, K))
                                   if prior else (None, None)))
    bq = np.zeros((C, K))
    for i in range(2):
        (qf, af, ebf, eaf, dtf), (qg, ag, ebg, eag, dtg) = got[True][0][i], got[False][0][i]
        assert np.array_equal(af, ag) and np.array_equal(dtf, dtg)
        if i == 0:
            assert np.array_equal(ebf, ebg)
        q_start = q0 if i == 0 else got[False][0][0][0]
        dt_used = np.full(C, dt) if i == 0 else got[False][0][0][4]
        for c in range(C):
            # both paths are within the bound of the numpy trajectory -> twice the
            # bound of each other
            b = pb.transition(q_start[c], p0[i][c], tau[c], dt_used[c], L, bq0=bq[c])
            assert abs(ebf[c] - ebg[c]) <= 2 * b['be_before']
            assert abs(eaf[c] - eag[c]) <= 2 * b['be_after']
            if af[c]:
                assert np.all(np.abs(qf[c] - qg[c]) <= 2 * b['bq'] + 4 * PB.U * np.abs(qg[c]))
                bq[c] = b['bq'] + 4 * PB.U * np.abs(qg[c])
    assert np.array_equal(got[True][1], got[False][1])


class _SmallConditional(R.PolyCoefficientsConditional):
    """The oracle's coefficient conditional with the component set / names of
    ``_small_posterior`` (no coefficients prior; another likelihood name changes
    the sorted summation order of the components)."""

    def __init__(self, *a, **kw):
        self.with_prior = kw.pop('with_prior')
        self.lik_name = kw.pop('lik_name')
        R.PolyCoefficientsConditional.__init__(self, *a, **kw)

    def component_log_probs(self, c):
        comps = R.PolyCoefficientsConditional.component_log_probs(self, c)
        comps[self.lik_name] = comps.pop('points')
        if not self.with_prior:
            del comps['coefficients_prior']
        return comps


@pytest.mark.parametrize('K,N,C,L,prior,lik_name', [
    (4, 20, 70, 50, True, 'points'),       # example_script.py's shape
    (4, 20, 5, 7, True, 'a_first'),
    (1, 1, 3, 3, False, 'points'),
    (7, 8, 9, 4, True, 'points'),
    (8, 37, 130, 5, True, 'points'),
    (9, 100, 6, 3, False, 'points'),
    (16, 128, 66, 2, True, 'points'),
    # more than 128 data points: one wave per chain (csrc/hmc_poly_wave.hip); regular and
    # ragged pairwise trees, several chains per wave, a partial last wave
    (4, 129, 9, 5, True, 'points'),
    (1, 200, 3, 3, False, 'points'),
    (8, 258, 70, 3, True, 'a_first'),
    (5, 512, 40, 4, True, 'points'),
    (16, 700, 5, 2, False, 'points'),
    (12, 1000, 17, 3, True, 'points'),
    (3, 1024, 33, 4, True, 'points')])
def test_fused_small_polynomial_transition_vs_oracle(device, K, N, C, L, prior, lik_name):
    """binf_hmc_sample_poly_f64 (one launch per transition) DIRECTLY against the
    numpy restatement -- RefHMCSampler on the coefficient conditional, one chain
    at a time, same injected draws -- for two consecutive transitions with
    per-chain precision and per-chain adapting step sizes: accept flags and step
    sizes identical, states and energies inside the computed bound."""
    rs = np.random.RandomState(100 * K + N + 1)
    xs = np.linspace(-1.5, 1.5, N)
    ys = R.polyval(xs, rs.standard_normal(K)) + 0.5 * rs.standard_normal(N)
    q0 = 0.2 * rs.standard_normal((C, K))
    p0 = rs.standard_normal((2, C, K))
    u = rs.uniform(size=(2, C))
    tau = 0.5 + rs.uniform(size=C)
    dt = 0.02 / max(1.0, N / 20.0) / K
    mu, var = np.linspace(-0.5, 0.5, K), np.linspace(2.0, 5.0, K)
    post = _small_posterior(xs, ys, K, prior, lik_name)
    cond = post.conditional_factory(precision=dev_t(tau, device))
    s = HMCSampler(cond, dev_t(q0, device), dt, L, timestep_adaption_limit=10,
                   variable_name='coefficients')
    assert s._fused_spec('coefficients', K) is not None and s.fused_transition
    got = []
    for i in range(2):
        out = s.sample(p0=dev_t(p0[i], device), u=dev_t(u[i], device))
        got.append((out.cpu().numpy().copy(), s.last_move_accepted.cpu().numpy().copy(),
                    s.last_e_before.cpu().numpy().copy(), s.last_e_after.cpu().numpy().copy(),
                    s.timestep.cpu().numpy().copy()))
    pb = PB.PolyBound(xs, ys, K, *((mu, var) if prior else (None, None)))
    n_acc = 0
    for c in range(C):
        draws = {'i': 0}
        pdf = _SmallConditional(xs, ys, tau[c], prior_means=mu, prior_variances=var,
                                gamma_shape=1.0, gamma_rate=1.0,     # quirk Q6: rate == shape
                                with_prior=prior, lik_name=lik_name)
        ref = R.RefHMCSampler(pdf, q0[c].copy(), dt, L, timestep_adaption_limit=10,
                              variable_name='coefficients',
                              normal=lambda size, c=c: p0[draws['i']][c].copy(),
                              uniform=lambda c=c: u[draws['i']][c])
        bq = np.zeros(K)
        q_start = q0[c]
        for i in range(2):
            draws['i'] = i
            dt_used = ref.timestep
            want = ref.sample()
            q, a, eb, ea, dts = got[i]
            b = pb.transition(q_start, p0[i][c], tau[c], dt_used, L, bq0=bq)
            assert bool(a[c]) == bool(ref.last_move_accepted), (c, i)
            assert dts[c] == ref.timestep, (c, i)
            assert abs(eb[c] - ref.last_E_before) <= b['be_before'], (c, i)
            assert abs(ea[c] - ref.last_E_after) <= b['be_after'], (c, i)
            if a[c]:
                bq = b['bq'] + 4 * PB.U * np.abs(want)
                n_acc += 1
            assert np.all(np.abs(q[c] - want) <= bq + 4 * PB.U * np.abs(want)), (c, i)
            q_start = want
    assert n_acc > 0


def test_fused_small_polynomial_limits_and_fallback(device):
    """More than 1024 data points (or a pairwise tree deeper than 3) or more than 16
    coefficients: the sampler stays on the per-step tier; the C entry point itself
    refuses."""
    rs = np.random.RandomState(0)
    for K, N in ((4, 1025), (4, 1023), (17, 20)):
        xs = np.linspace(-1, 1, N)
        ys = rs.standard_normal(N)
        cond = _small_posterior(xs, ys, K, True).conditional_factory(precision=2.0)
        s = HMCSampler(cond, dev_t(0.1 * rs.standard_normal((3, K)), device), 1e-3, 2,
                       variable_name='coefficients')
        assert s._fused_spec('coefficients', K) is None
        assert s.sample().shape == (3, K)
        z = torch.zeros((3, K), dtype=torch.float64, device=device)
        with pytest.raises(NotImplementedError):
            _native.hmc_sample_poly(z, z.clone(), torch.zeros(3, dtype=torch.float64, device=device),
                                    z.clone(), torch.zeros(3, dtype=torch.uint8, device=device),
                                    None, None, None, dev_t(xs, device), dev_t(ys, device), 2.0,
                                    None, None, True, None, None, 1e-3, None, 2, False,
                                    1.05, 0.95)
    # medium data sets: fused while the batch is launch-bound, per-step tier beyond
    xs = np.linspace(-1, 1, 1024)
    cond = _small_posterior(xs, rs.standard_normal(1024), 16, True).conditional_factory(precision=2.0)
    s = HMCSampler(cond, dev_t(np.zeros((3, 16)), device), 1e-4, 2, variable_name='coefficients')
    assert s._fused_spec('coefficients', 16, 1024) is not None
    assert s._fused_spec('coefficients', 16, 8192) is not None
    assert s._fused_spec('coefficients', 16, 65536) is None
    s.fused_transition = 'always'
    assert s._fused_spec('coefficients', 16, 65536) is not None
    # a posterior with another free variable is not the conditional the kernel integrates
    full = _small_posterior(np.linspace(-1, 1, 20), rs.standard_normal(20), 4, True)
    assert full.native_hmc_spec('coefficients') is None


def test_gamma_sampler_matches_restatement(device):
    xs, ys = RE.example_data()
    K, C = 4, 10
    rs = np.random.RandomState(4)
    theta = rs.standard_normal((C, K))
    post = make_posterior(xs, ys, POLYVAL)
    cond = post.conditional_factory(coefficients=dev_t(theta, device))
    g = rs.gamma(R.gamma_shape(len(ys), 1.0), size=C)
    gs = GammaSampler(cond, None, gamma=lambda shape, n, dev: dev_t(g, dev))
    assert gs._calculate_shape() == R.gamma_shape(len(ys), 1.0) == 10.0
    assert gs._get_prior().rate == 1.0                      # quirk Q6
    tau = gs.sample().cpu().numpy()
    want = np.array([R.gamma_draw(g[c], R.gamma_rate(xs, ys, theta[c], 1.0))
                     for c in range(C)])
    assert np.array_equal(tau, want)
    # default source: the global legacy stream, np.random.gamma(shape, size=C)
    np.random.seed(7)
    g2 = np.random.gamma(10.0, size=C)
    np.random.seed(7)
    tau2 = GammaSampler(cond, None).sample().cpu().numpy()
    assert np.array_equal(tau2, np.array([g2[c] / R.gamma_rate(xs, ys, theta[c], 1.0)
                                          for c in range(C)]))


def test_gibbs_within_hmc_c1_plumbing(device):
    """BASELINE config C1: example_script.py's posterior, HMC (50 leapfrog
    steps) on the coefficients inside Gibbs, precision by the conjugate
    update; several sweeps of several chains against the single-chain
    restatement with the same injected draws."""
    xs, ys = RE.example_data()
    K, C, L, dt, S = 4, 8, 50, 0.02, 4
    rs = np.random.RandomState(21)
    p0 = rs.standard_normal((S, C, K))
    u = rs.uniform(size=(S, C))
    shape = R.gamma_shape(len(ys), 1.0)
    g = rs.gamma(shape, size=(S, C))
    coeffs0 = np.ones((C, K)) + 0.05 * rs.standard_normal((C, K))
    tau0 = np.ones(C)

    start = BinfState(dict(coefficients=dev_t(coeffs0, device),
                           precision=dev_t(tau0, device)))
    post = make_posterior(xs, ys, POLYVAL)
    sweep = {'s': 0}
    gips = make_hmc_sampler(post, dt, L, start,
                            gamma=lambda sh, n, d: dev_t(g[sweep['s']], d))
    hmc = gips.subsamplers['coefficients']
    got_c, got_t, got_a = [], [], []
    for s in range(S):
        sweep['s'] = s
        hmc.rng = type('Inject', (), {
            'normal': staticmethod(lambda shp, d, s=s: dev_t(p0[s], d)),
            'uniform': staticmethod(lambda n, d, s=s: dev_t(u[s], d))})()
        state = gips.sample()
        got_c.append(state.variables['coefficients'].cpu().numpy().copy())
        got_t.append(state.variables['precision'].cpu().numpy().copy())
        got_a.append(hmc.last_move_accepted.cpu().numpy().copy())
    stats = gips.last_draw_stats
    assert set(stats) == {'coefficients'} and stats['coefficients'].stepsize == dt
    pb = PB.PolyBound(xs, ys, K, np.zeros(K), np.ones(K) * 5)
    for c in range(C):
        ref = RE.gibbs_hmc_chain(xs, ys, coeffs0[c], tau0[c], dt, L,
                                 p0[:, c], u[:, c], g[:, c])
        bounds = PB.gibbs_bounds(pb, ref['coefficients'], ref['accepted'], p0[:, c],
                                 ref['precision'], tau0[c], coeffs0[c], dt, L,
                                 RE.PRIOR_RATE_IN_CONDITIONALS)
        for s in range(S):
            assert bool(got_a[s][c]) == bool(ref['accepted'][s]), (c, s)
            assert np.all(np.abs(got_c[s][c] - ref['coefficients'][s]) <=
                          bounds[s]['bq'] + 4 * PB.U * np.abs(ref['coefficients'][s])), (c, s)
            assert abs(got_t[s][c] - ref['precision'][s]) <= \
                bounds[s]['btau'] * ref['precision'][s], (c, s)


def test_rwmc_gibbs_factory_runs_and_accepts(device):
    """The reference's own make_sampler wiring (RWMC + Gamma), batched."""
    xs, ys = RE.example_data()
    C = 32
    start = BinfState(dict(coefficients=torch.ones((C, 4), dtype=torch.float64, device=device),
                           precision=torch.ones(C, dtype=torch.float64, device=device)))
    np.random.seed(3)
    gips = make_sampler(make_posterior(xs, ys, POLYVAL), 0.1, start)
    for _ in range(30):
        state = gips.sample()
    rate = gips.last_draw_stats['coefficients'].acceptance_rate
    assert rate.shape == (C,) and 0.0 < float(rate.mean()) < 1.0
    assert torch.isfinite(state.variables['coefficients']).all()
    assert (state.variables['precision'] > 0).all()


def test_c3_full_size_gradient_properties(device):
    """BASELINE C3 size (K=33, N=16384, C=8192): linearity of the force in
    theta (the model is linear-Gaussian) and agreement with numpy on a sample
    of chains."""
    K, N, C = 33, 16384, 8192
    xs, ys, _ = synth(K, N, 1, 7)
    rs = np.random.RandomState(8)
    theta = rs.standard_normal((C, K))
    fwm = ForwardModel(xs, POLYVAL)
    A, tys = fwm.design_matrix(K, device), dev_t(ys, device)
    g1 = _native.poly_gauss_grad(dev_t(theta, device), A, tys, 2.5)
    g0 = _native.poly_gauss_grad(torch.zeros((C, K), dtype=torch.float64, device=device), A, tys, 2.5)
    g2 = _native.poly_gauss_grad(dev_t(2.0 * theta, device), A, tys, 2.5)
    # g(2 theta) - g(0) == 2 (g(theta) - g(0)) up to the rounding of the three
    # contractions, each within 1e-10 of its sum-of-magnitudes scale B
    lhs, rhs = (g2 - g0).cpu().numpy(), 2.0 * (g1 - g0).cpu().numpy()
    Jn = np.vstack([xs ** i for i in range(K)])
    aJ = np.abs(Jn)
    for c in (0, 17, 4095, 8191):
        r0 = (0.0 - ys) * 2.5
        r1 = (POLYVAL(xs, theta[c]) - ys) * 2.5
        r2 = (POLYVAL(xs, 2.0 * theta[c]) - ys) * 2.5
        B0, B1, B2 = aJ.dot(np.abs(r0)), aJ.dot(np.abs(r1)), aJ.dot(np.abs(r2))
        assert np.all(np.abs(lhs[c] - rhs[c]) <= RTOL * (B2 + 3 * B0 + 2 * B1))
        want = Jn.dot(r1)
        assert np.all(np.abs(g1[c].cpu().numpy() - want) <= RTOL * B1)
    lp = _native.poly_gauss_logp(dev_t(theta, device), dev_t(xs, device), tys, 1.0).cpu().numpy()
    for c in (0, 8191):
        assert lp[c] == -0.5 * np.sum((POLYVAL(xs, theta[c]) - ys) ** 2) * 1.0 + N * 0.5 * np.log(1.0)


def test_example_script_counterpart_recovers_the_coefficients(device):
    """examples/polynomial_fit.py: the reference's example_script.py flow with
    many chains; the posterior mean must sit near the least-squares fit."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        'examples', 'polynomial_fit.py')
    spec = importlib.util.spec_from_file_location('polynomial_fit', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    coeffs, prec = mod.main(['--chains', '256', '--iterations', '300', '--burn-in', '150',
                             '--thin', '10', '--seed', '0'])
    assert coeffs.shape == (15, 256, 4) and prec.shape == (15, 256, 1)
    xs, ys = RE.example_data()
    lsq = np.polynomial.polynomial.polyfit(xs, ys, 3)
    mean = coeffs.reshape(-1, 4).mean(0).cpu().numpy()
    assert np.abs(mean - lsq).max() < 0.25
    assert 0.5 < float(prec.mean()) < 6.0


@pytest.mark.parametrize('path', golden_files('poly_'))
def test_gibbs_within_hmc_reproduces_golden_vectors(device, path):
    """Committed fixtures (tests/golden/poly_*.npz: Gibbs sweeps of several
    chains, draws recorded from the reference's np.random consumption order;
    poly_c1_example is example_script.py's own data and start) through the
    class stack: Posterior -> Gibbs -> fused HMC transition + conjugate update."""
    g = load_golden(path)
    K, L, dt = int(g['K']), int(g['L']), float(g['timestep'])
    S, C = g['u'].shape
    from binf_amd.pdf.likelihoods import Likelihood
    from binf_amd.pdf.posteriors import Posterior
    lik = Likelihood('points', ForwardModel(g['xs'], POLYVAL), GaussianErrorModel(g['ys']))
    post = Posterior({lik.name: lik}, {'precision_prior': GammaPrior(1.0, 0.2),
                                       'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K) * 5)})
    start = BinfState(dict(coefficients=dev_t(g['coefficients0'], device),
                           precision=dev_t(g['precision0'], device)))
    sweep = {'s': 0}
    gips = make_hmc_sampler(post, dt, L, start,
                            gamma=lambda sh, n, d: dev_t(g['gamma'][sweep['s']], d))
    hmc = gips.subsamplers['coefficients']
    assert gips.subsamplers['precision']._calculate_shape() == float(g['gamma_shape'])
    pb = PB.PolyBound(g['xs'], g['ys'], K, np.zeros(K), np.ones(K) * 5)
    bounds = [PB.gibbs_bounds(pb, g['coefficients'][:, c], g['accepted'][:, c], g['p0'][:, c],
                              g['precision'][:, c], g['precision0'][c], g['coefficients0'][c],
                              dt, L, RE.PRIOR_RATE_IN_CONDITIONALS) for c in range(C)]
    for s in range(S):
        sweep['s'] = s
        hmc.rng = type('Inject', (), {
            'normal': staticmethod(lambda shp, d, s=s: dev_t(g['p0'][s], d)),
            'uniform': staticmethod(lambda n, d, s=s: dev_t(g['u'][s], d))})()
        state = gips.sample()
        assert np.array_equal(hmc.last_move_accepted.cpu().numpy(), g['accepted'][s].astype(bool)), s
        c = state.variables['coefficients'].cpu().numpy()
        t = state.variables['precision'].cpu().numpy()
        eb, ea = hmc.last_e_before.cpu().numpy(), hmc.last_e_after.cpu().numpy()
        for k in range(C):
            b = bounds[k][s]
            want = g['coefficients'][s][k]
            assert np.all(np.abs(c[k] - want) <= b['bq'] + 4 * PB.U * np.abs(want)), (s, k)
            assert abs(t[k] - g['precision'][s][k]) <= b['btau'] * g['precision'][s][k], (s, k)
            assert abs(eb[k] - g['e_before'][s][k]) <= b['be_before'], (s, k)
            assert abs(ea[k] - g['e_after'][s][k]) <= b['be_after'], (s, k)


def test_gradient_calls_on_two_streams_do_not_share_scratch(device):
    """The split-data gradient reduces through a cached scratch buffer; two
    streams running it concurrently must each get their own."""
    K, N, C = 33, 16384, 2048
    xs, ys, theta = synth(K, N, C, 3)
    A = dev_t(np.vstack([xs ** i for i in range(K)]), device)
    tys = dev_t(ys, device)
    th1, th2 = dev_t(theta, device), dev_t(theta[::-1].copy(), device)
    want1 = _native.poly_gauss_grad(th1, A, tys, 2.5).clone()
    want2 = _native.poly_gauss_grad(th2, A, tys, 2.5).clone()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(device), torch.cuda.Stream(device)
    for _ in range(5):
        with torch.cuda.stream(s1):
            g1 = _native.poly_gauss_grad(th1, A, tys, 2.5)
        with torch.cuda.stream(s2):
            g2 = _native.poly_gauss_grad(th2, A, tys, 2.5)
        torch.cuda.synchronize()
        assert torch.equal(g1, want1) and torch.equal(g2, want2)


@pytest.mark.parametrize('seed', [0, 1, 7])
def test_example_script_itself_one_chain_same_stream(device, seed):
    """example_script.py as the reference ships it -- ONE chain, RWMC + Gamma
    inside Gibbs (make_sampler), every draw from the global legacy np.random
    stream that also generated the data -- against its numpy restatement: the
    chain-batched samplers with C = 1 consume the stream in the same order, and
    every state of the chain must match bit for bit."""
    sweeps = 300
    ref = RE.example_script_chain(seed, sweeps)
    np.random.seed(seed)
    xs = np.linspace(-2, 2, 20)
    ys = np.random.normal(loc=R.polyval(xs, np.array([2.0, -4.0, 1.0, 1.5])),
                          scale=1.0 / np.sqrt(2.5))
    assert np.array_equal(ys, ref['ys'])
    start = BinfState(dict(coefficients=dev_t(np.ones((1, 4)), device),
                           precision=dev_t(np.ones(1), device)))
    gips = make_sampler(make_posterior(xs, ys, POLYVAL), 0.1, start)
    for s in range(sweeps):
        st = gips.sample()
        c = st.variables['coefficients'].cpu().numpy()[0]
        t = float(st.variables['precision'].cpu().numpy()[0])
        assert np.array_equal(c, ref['coefficients'][s]), s
        assert t == ref['precision'][s], s
    rate = gips.last_draw_stats['coefficients'].acceptance_rate
    assert abs(float(rate) - ref['acceptance_rate']) < 1e-12
    assert 0.05 < ref['acceptance_rate'] < 0.95


def test_gradient_summation_order_depends_on_the_batch_only_within_the_bar(device):
    """The split of the data range follows (C, N): the same chain in batches of
    different size may differ at rounding level (documented in the ABI header and
    DESIGN.md section 6) -- pinned here: always inside 1e-10 of the
    sum-of-magnitudes scale, bit-identical when the workspace size says the split
    is the same, and a missing workspace is an error, not another order."""
    K, N = 33, 16384
    xs, ys, theta = synth(K, N, 4096, 5)
    A = ForwardModel(xs, POLYVAL).design_matrix(K, device)
    tys = dev_t(ys, device)
    full = _native.poly_gauss_grad(dev_t(theta, device), A, tys, 2.5).cpu().numpy()
    Jn = np.vstack([xs ** i for i in range(K)])
    lib = _native.lib()
    for Cs in (1, 64, 1000, 2048):
        part = _native.poly_gauss_grad(dev_t(theta[:Cs], device), A, tys, 2.5).cpu().numpy()
        for c in (0, Cs - 1):
            B = np.abs(Jn).dot(np.abs((POLYVAL(xs, theta[c]) - ys) * 2.5))
            assert np.all(np.abs(part[c] - full[c]) <= RTOL * B)
    # same chains, same batch size, another position in the batch: same bits
    a = _native.poly_gauss_grad(dev_t(theta[:2048], device), A, tys, 2.5)
    b = _native.poly_gauss_grad(dev_t(theta[1024:3072], device), A, tys, 2.5)
    assert torch.equal(a[1024:], b[:1024])
    need = lib.binf_poly_gauss_grad_workspace_bytes(64, K, N)
    assert need > 0
    out = torch.empty((64, K), dtype=torch.float64, device=device)
    th = dev_t(theta[:64], device)
    rc = lib.binf_poly_gauss_grad_f64(th.data_ptr(), A.data_ptr(), tys.data_ptr(), 2.5, None,
                                      out.data_ptr(), None, 0, 64, K, N,
                                      _native.stream_handle(device))
    assert rc == _native.E_ARG and 'workspace' in _native.last_error()


def test_more_than_65535_chains_per_call(device):
    """gridDim.y chunks: forward model and error-model gradient for a batch larger
    than one grid dimension."""
    K, N, C = 4, 20, 70000
    xs, ys, _ = synth(K, N, 1, 3, xlim=2.0)
    theta = np.random.RandomState(0).standard_normal((C, K))
    mock = _native.poly_forward(dev_t(theta, device), dev_t(xs, device))
    for c in (0, 65534, 65535, 65536, C - 1):
        assert np.array_equal(mock[c].cpu().numpy(), POLYVAL(xs, theta[c]))
    taus = np.random.RandomState(1).uniform(0.5, 2.0, size=C)
    g = _native.gauss_err_grad(mock, dev_t(ys, device), dev_t(taus, device))
    for c in (0, 65535, C - 1):
        assert np.array_equal(g[c].cpu().numpy(), (POLYVAL(xs, theta[c]) - ys) * taus[c])


def test_subclassed_models_are_evaluated_as_written_and_new_data_is_seen(device):
    """A subclass that overrides an evaluation method must not be routed to the
    fused kernels of its base class; reassigning the model data drops the cached
    device copies."""
    class Shifted(ForwardModel):
        def _evaluate(self, coefficients):
            return ForwardModel._evaluate(self, coefficients) + 1.0

    class Scaled(GaussianErrorModel):
        def _evaluate_gradient(self, mock_data, precision):
            return 2.0 * GaussianErrorModel._evaluate_gradient(self, mock_data, precision)

    xs, ys, theta = synth(4, 20, 5, 1, xlim=2.0)
    assert ForwardModel(xs, POLYVAL).native_spec() is not None
    assert Shifted(xs, POLYVAL).native_spec() is None
    assert GaussianErrorModel(ys).native_spec() is not None
    assert Scaled(ys).native_spec() is None
    from binf_amd.pdf.likelihoods import Likelihood
    tc = dev_t(theta, device)
    plain = Likelihood('points', ForwardModel(xs, POLYVAL), GaussianErrorModel(ys))
    shifted = Likelihood('points', Shifted(xs, POLYVAL), GaussianErrorModel(ys))
    lp0 = plain.log_prob(coefficients=tc, precision=1.0).cpu().numpy()
    lp1 = shifted.log_prob(coefficients=tc, precision=1.0).cpu().numpy()
    for c in range(5):
        mock = POLYVAL(xs, theta[c])
        assert lp0[c] == -0.5 * np.sum((mock - ys) ** 2) * 1.0
        assert abs(lp1[c] - (-0.5 * np.sum((mock + 1.0 - ys) ** 2))) <= 1e-12 * abs(lp1[c])
    # new data on the same model object
    em = plain.error_model
    em.ys = ys + 1.0
    lp2 = plain.log_prob(coefficients=tc, precision=1.0).cpu().numpy()
    assert lp2[0] == -0.5 * np.sum((POLYVAL(xs, theta[0]) - (ys + 1.0)) ** 2) * 1.0
    fm = plain.forward_model
    fm.xses = xs * 0.5
    lp3 = plain.log_prob(coefficients=tc, precision=1.0).cpu().numpy()
    assert lp3[0] == -0.5 * np.sum((POLYVAL(xs * 0.5, theta[0]) - (ys + 1.0)) ** 2) * 1.0


def test_c4_full_size_gibbs_sweep(device):
    """BASELINE C4 at its per-GPU size (K = 33, N = 16384, 4096 chains = 32768 / 8):
    two Gibbs sweeps (HMC with L = 20 on the coefficients, conjugate precision
    update) through the class stack with injected draws; a sample of chains against
    the single-chain numpy restatement inside the computed bounds, accept flags
    identical, everything finite."""
    K, N, C, L, dt, S = 33, 16384, 4096, 20, 2e-4, 2
    xs, ys, _ = synth(K, N, 1, 7)
    rs = np.random.RandomState(12)
    c0 = 0.3 * rs.standard_normal((C, K))
    tau0 = rs.uniform(1.5, 3.5, size=C)
    p0 = rs.standard_normal((S, C, K))
    u = rs.uniform(size=(S, C))
    shape = R.gamma_shape(N, 1.0)
    g = rs.gamma(shape, size=(S, C))
    from binf_amd.pdf.likelihoods import Likelihood
    from binf_amd.pdf.posteriors import Posterior
    lik = Likelihood('points', ForwardModel(xs, POLYVAL), GaussianErrorModel(ys))
    post = Posterior({lik.name: lik}, {'precision_prior': GammaPrior(1.0, 0.2),
                                       'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K) * 5)})
    start = BinfState(dict(coefficients=dev_t(c0, device), precision=dev_t(tau0, device)))
    sweep = {'s': 0}
    gips = make_hmc_sampler(post, dt, L, start, gamma=lambda sh, n, d: dev_t(g[sweep['s']], d))
    hmc = gips.subsamplers['coefficients']
    got = []
    for s in range(S):
        sweep['s'] = s
        hmc.rng = type('Inject', (), {
            'normal': staticmethod(lambda shp, d, s=s: dev_t(p0[s], d)),
            'uniform': staticmethod(lambda n, d, s=s: dev_t(u[s], d))})()
        st = gips.sample()
        got.append((st.variables['coefficients'].cpu().numpy().copy(),
                    st.variables['precision'].cpu().numpy().copy(),
                    hmc.last_move_accepted.cpu().numpy().copy()))
    for cs, ts, acc in got:
        assert np.isfinite(cs).all() and np.isfinite(ts).all() and (ts > 0).all()
    assert got[-1][2].mean() > 0.5
    pb = PB.PolyBound(xs, ys, K, np.zeros(K), np.ones(K) * 5)
    for c in (0, 4095):
        ref = RE.gibbs_hmc_chain(xs, ys, c0[c], tau0[c], dt, L, p0[:, c], u[:, c], g[:, c])
        bounds = PB.gibbs_bounds(pb, ref['coefficients'], ref['accepted'], p0[:, c],
                                 ref['precision'], tau0[c], c0[c], dt, L,
                                 RE.PRIOR_RATE_IN_CONDITIONALS)
        for s in range(S):
            assert bool(got[s][2][c]) == bool(ref['accepted'][s]), (c, s)
            want = ref['coefficients'][s]
            assert np.all(np.abs(got[s][0][c] - want) <= bounds[s]['bq'] + 4 * PB.U * np.abs(want)), (c, s)
            assert abs(got[s][1][c] - ref['precision'][s]) <= bounds[s]['btau'] * ref['precision'][s], (c, s)


def test_gamma_prior_log_prob_kernel_equals_the_expression(device):
    """binf_gamma_logp_f64 against the four-operation expression of
    binf/example/priors.py:10-25 -- bit for bit vs the torch ops it replaces, and
    within an ulp-level tolerance of numpy (whose log is a different libm)."""
    from binf_amd.example.priors import GammaPrior
    rs = np.random.RandomState(4)
    for C in (1, 5, 257, 70001):
        tau = np.concatenate([rs.uniform(1e-3, 50.0, size=C - 1), [1.0]]) if C > 1 else np.array([2.5])
        t = dev_t(tau, device)
        for shape, rate in ((1.0, 0.2), (3.5, 3.5), (0.3, 7.0)):
            got = _native.gamma_logp(t, shape, rate)
            want_t = (shape - 1.0) * torch.log(t) - t * rate
            assert torch.equal(got, want_t)
            want = (shape - 1.0) * np.log(tau) - tau * rate
            assert np.allclose(got.cpu().numpy(), want, rtol=0, atol=4e-16 * np.maximum(1.0, np.abs(want)).max() * 8)
            pr = GammaPrior(shape, rate)
            assert torch.equal(pr.log_prob(precision=t), got)


@pytest.mark.parametrize('K,N,C,L,mode,per_chain', [
    (4, 20, 5, 3, 'exact', False), (33, 1000, 20, 4, 'exact', True), (33, 16384, 130, 2, 'exact', False),
    (8, 300, 2100, 3, 'fma', True), (17, 50, 3, 1, 'exact', False), (64, 129, 70, 2, 'fma', False),
    (33, 4096, 2050, 2, 'exact', True)])
def test_fused_transition_leapfrog_is_bit_identical_to_the_per_step_tier(device, K, N, C, L, mode,
                                                                         per_chain):
    """binf_poly_leapfrog_f64 (gradient + partial-sum reduction + kick + drift per
    launch, the last workgroup of a chain tile combining it) against the per-step
    sequence of launches it replaces: the same bits in q and p, whatever workgroup
    happened to come last."""
    xs, ys, theta = synth(K, N, C, K + N, xlim=1.0)
    lik = make_likelihood(xs, ys, POLYVAL)
    post = Posterior({lik.name: lik}, {'precision_prior': GammaPrior(1.0, 0.2),
                                       'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K) * 5)})
    rs = np.random.RandomState(C)
    taus = dev_t(rs.uniform(1.0, 4.0, size=C), device)
    cond = post.conditional_factory(precision=taus)
    spec = cond.native_leapfrog_spec('coefficients')
    assert spec is not None and spec[0] == 'poly'
    assert cond.native_leapfrog_spec('precision') is None
    dt = 1e-3 / K
    dts = dev_t(dt * rs.uniform(0.5, 1.5, size=C), device) if per_chain else dt
    p0 = rs.standard_normal((C, K))
    outs = []
    for fused in (True, False, True):
        s = HMCSampler(cond, dev_t(theta, device), dt, L, variable_name='coefficients', mode=mode)
        s.fused_leapfrog = fused
        s.fused_transition = False
        q, p = dev_t(theta, device), dev_t(p0, device)
        s._leapfrog(q, p, dts, L)
        outs.append((q, p))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    assert not torch.equal(outs[0][0], dev_t(theta, device))
    # ... and a whole sample() through it equals the per-step one
    res = []
    for fused in (True, False):
        s = HMCSampler(cond, dev_t(theta, device), dt, L, variable_name='coefficients', mode=mode,
                       record_energies=True)
        s.fused_leapfrog = fused
        s.fused_transition = False
        x = s.sample(p0=dev_t(p0, device), u=dev_t(rs.uniform(size=C) * 0 + 0.5, device))
        res.append((x, s.last_e_after, s.last_move_accepted))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_fused_transition_leapfrog_argument_checks(device):
    C, K, N = 8, 4, 20
    q = torch.zeros((C, K), dtype=torch.float64, device=device)
    A = torch.ones((K, N), dtype=torch.float64, device=device)
    y = torch.zeros(N, dtype=torch.float64, device=device)
    with pytest.raises(ValueError):
        _native.poly_leapfrog(q, q.clone(), A, y, 1.0, 0.1, None, 0)
    with pytest.raises(ValueError):
        _native.poly_leapfrog(q, q.clone(), A[:, :5].contiguous(), y, 1.0, 0.1, None, 2)
    need = _native.lib().binf_poly_leapfrog_workspace_bytes(C, K, N)
    assert need > 0
    rc = _native.lib().binf_poly_leapfrog_f64(q.data_ptr(), q.clone().data_ptr(), A.data_ptr(),
                                              y.data_ptr(), 1.0, None, None, 0, C, K, N, 0.1, None, 2,
                                              0, None)
    assert rc == _native.E_ARG and 'workspace' in _native.last_error()


def test_chi2_memo_is_checked_on_the_device_chain_by_chain(device):
    """binf_poly_gauss_logp_memo_f64: whatever happens to the coefficient buffers in
    between (in-place kernel writes, reallocation at the same address, partial
    changes), the result is the plain log-prob bit for bit; chains whose
    coefficients are unchanged are not summed again (skip flags)."""
    K, N, C = 33, 4096, 37
    xs, ys, theta = synth(K, N, C, 5)
    tx, ty = dev_t(xs, device), dev_t(ys, device)
    nan = float('nan')
    memo = _native.new_chi2_memo(C, K, device)
    reused = lambda: memo[2][0].cpu().numpy().astype(bool)
    rs = np.random.RandomState(0)
    th = dev_t(theta, device)
    taus = dev_t(rs.uniform(0.5, 3.0, size=C), device)
    changed_prev = np.ones(C, dtype=bool)
    for step in range(6):
        prec = (2.5, 1.0, taus)[step % 3]
        got = _native.poly_gauss_logp_memo(th, tx, ty, prec, memo)
        want = _native.poly_gauss_logp(th, tx, ty, prec)
        assert np.array_equal(got.cpu().numpy(), want.cpu().numpy(), equal_nan=True), step
        assert np.array_equal(reused(), ~changed_prev), step
        entry = memo[0][memo[2][1].long(), torch.arange(C, device=device)]     # the entry each chain used
        assert np.array_equal(entry.cpu().numpy(), th.cpu().numpy(), equal_nan=True)
        # change a random subset IN PLACE (as a kernel would: no new tensor, no version bump
        # that anything here looks at), including a sign flip of a zero and a NaN
        changed_prev = rs.rand(C) < 0.4
        idx = np.nonzero(changed_prev)[0]
        upd = theta[idx] + rs.standard_normal((len(idx), K)) * 1e-3
        if step == 2 and len(idx) > 1:
            upd[0, 3] = nan
        theta[idx] = upd
        th[torch.from_numpy(idx).to(device)] = dev_t(upd, device)
    # +0.0 -> -0.0 is a different bit pattern: not reused (and the result is still exact)
    th[0, 0] = 0.0
    _native.poly_gauss_logp_memo(th, tx, ty, 2.5, memo)
    th[0, 0] = -0.0
    got = _native.poly_gauss_logp_memo(th, tx, ty, 2.5, memo)
    assert not reused()[0] and int(reused()[1:].sum()) == C - 1
    assert np.array_equal(got.cpu().numpy(), _native.poly_gauss_logp(th, tx, ty, 2.5).cpu().numpy(),
                          equal_nan=True)
    assert int(torch.isnan(got).sum()) == 1          # the chain that was given a NaN coefficient
    # two entries per chain: state, proposal, then EITHER of them again (the acceptance test
    # of hmc.py:152-158 went one way for some chains, the other way for the rest) is not
    # summed again -- and that for every transition that follows
    state = th.clone()
    for step in range(4):
        prop = state + 1e-3 * torch.randn_like(state)
        _native.poly_gauss_logp_memo(state, tx, ty, 2.5, memo)           # E_before
        assert reused().all() or step == 0
        _native.poly_gauss_logp_memo(prop, tx, ty, 2.5, memo)            # E_after
        assert not reused().any()
        acc = torch.from_numpy(rs.rand(C) < 0.5).to(device)
        state = torch.where(acc[:, None], prop, state)
        got = _native.poly_gauss_logp_memo(state, tx, ty, 1.0, memo)     # the precision update's chi^2
        assert reused().all()
        assert np.array_equal(got.cpu().numpy(), _native.poly_gauss_logp(state, tx, ty, 1.0).cpu().numpy(),
                              equal_nan=True)


def test_gibbs_sweeps_with_and_without_the_chi2_memo_are_identical(device):
    """Through the class stack (Posterior -> Likelihood -> native log-prob) at a data
    size that uses the memo: Gibbs-within-HMC sweeps with per-chain precisions give
    the same states with the memo switched off."""
    from binf_amd.example import native_poly
    K, N, C, L = 9, 2500, 21, 3
    xs, ys, theta = synth(K, N, C, 8)
    runs = []
    for use in (True, False, True):
        native_poly.USE_CHI2_MEMO = use
        try:
            lik = make_likelihood(xs, ys, POLYVAL)
            post = Posterior({lik.name: lik}, {'precision_prior': GammaPrior(1.0, 0.2),
                                               'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K) * 5)})
            st = BinfState(dict(coefficients=dev_t(theta, device),
                                precision=torch.full((C,), 2.0, dtype=torch.float64, device=device)))
            from binf_amd.samplers.rng import DeviceRNG
            gips = make_hmc_sampler(post, 2e-3, L, st, rng=DeviceRNG(3, device))
            out = []
            for _ in range(5):
                s = gips.sample()
                out.append((s.variables['coefficients'].clone(), s.variables['precision'].clone()))
            runs.append(out)
            acc = gips.subsamplers['coefficients'].acceptance_rate
            assert 0.0 < float(acc.mean()) <= 1.0
        finally:
            native_poly.USE_CHI2_MEMO = True
    for a, b, c in zip(*runs):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])

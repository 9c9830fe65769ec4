"""
Independent evidence for the parity-unpinned HMC numerics (CPU only).

The reference's ``binf/samplers/hmc.py:92-164`` is executed by none of its own
tests and cannot be imported here (csb absent), so the golden fixtures under
``tests/golden/`` are outputs of this build's restatement (``oracle/ref_numpy.py``):
**parity unpinned**, and this file does not change that status.  What it adds is
a check that shares NO code with ``oracle/``: the leapfrog recurrence
(``hmc.py:116-123``), the two energies (``hmc.py:148,150``) and the acceptance
ratio (``hmc.py:151``) are written here a second time, straight from those
lines, in 50-digit ``mpmath`` arithmetic, for

* the isotropic Gaussian in the reference's ``TestHO`` form
  (``binf/pdf/__init__.py:185,191``: ``log p = -0.5 k sum (x-x0)^2``,
  ``gradient = k (x-x0)``), and
* the example's conditional posterior of the polynomial coefficients
  (``binf/example/likelihood.py:24-30,54-61``, ``binf/example/priors.py:23-25,49-54``,
  force = the likelihood alone, ``binf/pdf/posteriors.py:183`` -- quirk Q4),

and every golden ``q_out`` / ``E_before`` / ``E_after`` has to lie within an
ulp-scaled bound of that exact trajectory, every golden accept decision has to
be the exact decision, and none may sit within 1e-12 of its ``u`` (a decision
that close could legitimately flip between two correct fp64 evaluations).
"""
import os

import mpmath
import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
EPS = float(np.finfo(np.float64).eps)
mpmath.mp.dps = 50
mpf = mpmath.mpf


def _mp(a):
    """float64 array -> object array of exact mpf values."""
    out = np.empty(np.shape(a), dtype=object)
    flat = out.reshape(-1)
    for i, v in enumerate(np.asarray(a, dtype=np.float64).reshape(-1)):
        flat[i] = mpf(float(v))
    return out


def _f(a):
    return np.array([float(v) for v in np.asarray(a, dtype=object).reshape(-1)]).reshape(np.shape(a))


def _leapfrog(q, p, dt, L, grad):
    # hmc.py:116-123, in exact arithmetic
    q = q.copy()
    p = p.copy()
    half = mpf(0.5) * dt
    p = p - half * grad(q)
    for _ in range(L - 1):
        q = q + p * dt
        p = p - dt * grad(q)
    q = q + p * dt
    p = p - half * grad(q)
    return q, p


def _sum(a):
    return mpmath.fsum(list(np.asarray(a, dtype=object).reshape(-1)))


def _check_transition(q0, p0, u, dt, L, V, grad, g_q, g_acc, g_eb, g_ea, scale_q, ctx,
                      c_state, c_energy):
    """One sample() (hmc.py:136-164) in exact arithmetic against its golden record."""
    q0m, p0m = _mp(q0), _mp(p0)
    dtm = mpf(float(dt))
    e_before = V(q0m) + mpf(0.5) * _sum(p0m * p0m)
    q1, p1 = _leapfrog(q0m, p0m, dtm, L, grad)
    e_after = V(q1) + mpf(0.5) * _sum(p1 * p1)
    ratio = mpmath.exp(-(e_after - e_before))
    acc = mpf(float(u)) < ratio
    # 1. the decision is the exact decision, with room
    assert bool(acc) == bool(g_acc), ctx
    assert abs(float(mpf(float(u)) - ratio)) > 1e-12, (ctx, float(u), float(ratio))
    # 2. energies: rounding of a D-term pairwise sum plus the propagated trajectory error
    mag_b = float(abs(V(q0m)) + mpf(0.5) * _sum(p0m * p0m))
    mag_a = float(abs(V(q1)) + mpf(0.5) * _sum(p1 * p1))
    tol_b = c_energy * EPS * max(mag_b, 1.0)
    tol_a = c_energy * (L + 2) * EPS * max(mag_a, 1.0)
    err_b = abs(float(mpf(float(g_eb)) - e_before))
    err_a = abs(float(mpf(float(g_ea)) - e_after))
    assert err_b <= tol_b, (ctx, 'E_before', err_b, tol_b)
    assert err_a <= tol_a, (ctx, 'E_after', err_a, tol_a)
    # 3. the state handed back (hmc.py:159-164)
    want = q1 if acc else q0m
    err_q = np.max(np.abs(_f(_mp(g_q) - want)))
    tol_q = (c_state * (L + 2) * EPS * scale_q) if acc else 0.0
    assert err_q <= tol_q, (ctx, 'q_out', err_q, tol_q)
    return err_q / max(tol_q, 1e-300), err_a / tol_a


GAUSS_SETS = ['gauss_d4_l1', 'gauss_d4_l50_k2p5', 'gauss_d7_l2', 'gauss_d33_l20',
              'gauss_d33_l2_adapt', 'gauss_d200_l20', 'gauss_d768_l50_k2p5',
              'gauss_d1024_l20', 'gauss_d1024_l20_bigdt', 'gauss_d1024_l1_adapt']


@pytest.mark.parametrize('name', GAUSS_SETS)
def test_gaussian_golden_trajectories_against_50_digit_arithmetic(name):
    d = np.load(os.path.join(GOLDEN, name + '.npz'))
    L, k, x0 = int(d['L']), mpf(float(d['k'])), mpf(float(d['x0']))
    T, C, D = d['p0'].shape
    # bound the cost: the large sets check their first chains only
    n_chains = C if D <= 256 else 2
    n_trans = T if D <= 256 else min(T, 2)

    def V(x):                      # -log_prob, binf/pdf/__init__.py:185
        return mpf(0.5) * k * _sum((x - x0) * (x - x0))

    def grad(x):                   # binf/pdf/__init__.py:191
        return k * (x - x0)

    worst = (0.0, 0.0)
    for c in range(n_chains):
        for t in range(n_trans):
            q0 = d['q0'][c] if t == 0 else d['q_out'][t - 1, c]
            dt = float(d['timestep']) if t == 0 else float(d['timestep_out'][t - 1, c])
            # amplitude of the (linear) motion: what an ulp of the trajectory is scaled by
            scale = float(np.max(np.abs(q0 - float(x0))) + np.max(np.abs(d['p0'][t, c]))
                          + abs(float(x0)))
            r = _check_transition(q0, d['p0'][t, c], d['u'][t, c], dt, L, V, grad,
                                  d['q_out'][t, c], d['accepted'][t, c],
                                  d['e_before'][t, c], d['e_after'][t, c], scale,
                                  (name, c, t), c_state=1.0, c_energy=1.0)
            worst = (max(worst[0], r[0]), max(worst[1], r[1]))
    # the bounds are not slack by orders of magnitude either
    assert worst[0] <= 1.0 and worst[1] <= 1.0


def test_adapted_timesteps_follow_hmc_py_183_191():
    # timestep *= uprate on ACCEPT, downrate on reject, while counter < limit after the
    # increment (hmc.py:153-157,188-191); exact products rounded once per step
    for name in ('gauss_d33_l2_adapt', 'gauss_d1024_l1_adapt', 'gauss_d8200_l2_adapt'):
        d = np.load(os.path.join(GOLDEN, name + '.npz'))
        T, C = d['u'].shape
        lim, up, down = int(d['adaption_limit']), float(d['uprate']), float(d['downrate'])
        for c in range(C):
            dt = float(d['timestep'])
            for t in range(T):
                if t + 1 < lim:
                    dt = dt * (up if d['accepted'][t, c] else down)
                assert dt == d['timestep_out'][t, c], (name, c, t)


@pytest.mark.parametrize('name', ['poly_c1_example', 'poly_k7_n37'])
def test_polynomial_conditional_golden_trajectories_against_50_digit_arithmetic(name):
    d = np.load(os.path.join(GOLDEN, name + '.npz'))
    K, N, L = int(d['K']), int(d['N']), int(d['L'])
    xs, ys = _mp(d['xs']), _mp(d['ys'])
    T, C, _ = d['p0'].shape
    # design matrix rows xs**i (likelihood.py:28-30), exact
    A = np.empty((K, N), dtype=object)
    for i in range(K):
        for n in range(N):
            A[i, n] = xs[n] ** i
    prior_var = mpf(5)             # priors.py:70: variances = ones * 5, means = 0
    prior_shape = mpf(1)           # GammaPrior(1.0, 0.2); the conditional copy has rate == shape (Q6)

    worst = (0.0, 0.0)
    for c in range(C):
        for t in range(T):
            th0 = d['coefficients0'][c] if t == 0 else d['coefficients'][t - 1, c]
            tau = mpf(float(d['precision0'][c] if t == 0 else d['precision'][t - 1, c]))

            def mock(theta):
                return np.array([_sum(theta * A[:, n]) for n in range(N)], dtype=object)

            def V(theta):
                r = mock(theta) - ys
                lik = -mpf(0.5) * _sum(r * r) * tau + mpf(N) * mpf(0.5) * mpmath.log(tau)
                cprior = -mpf(0.5) * _sum(theta * theta / prior_var)
                pprior = (prior_shape - 1) * mpmath.log(tau) - tau * prior_shape
                return -(lik + cprior + pprior)

            def grad(theta):       # likelihoods.py:148-155; the priors are not in the force (Q4)
                r = (mock(theta) - ys) * tau
                return np.array([_sum(A[i] * r) for i in range(K)], dtype=object)

            # an ulp of this trajectory: the force is J.r, conditioned like sum|J||r|
            r0 = np.abs(_f((mock(_mp(th0)) - ys) * tau))
            fscale = float(np.max(np.abs(_f(A)) @ r0))
            scale = float(np.max(np.abs(th0)) + np.max(np.abs(d['p0'][t, c]))
                          + float(d['timestep']) * L * fscale * float(d['timestep']))
            r = _check_transition(th0, d['p0'][t, c], d['u'][t, c], float(d['timestep']), L,
                                  V, grad, d['coefficients'][t, c], d['accepted'][t, c],
                                  d['e_before'][t, c], d['e_after'][t, c], scale,
                                  (name, c, t), c_state=1.0, c_energy=4.0)
            worst = (max(worst[0], r[0]), max(worst[1], r[1]))
    assert worst[0] <= 1.0 and worst[1] <= 1.0

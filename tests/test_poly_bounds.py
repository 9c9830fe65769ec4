"""The computed tolerance of tests/poly_bounds.py is itself tested on the CPU:
trajectories whose force differs from the oracle's in a KNOWN way (another
summation order; an injected error of exactly EPS * B) must stay inside the
propagated bound, and the bound must not be vacuous."""
import numpy as np
import pytest

import poly_bounds as PB
from oracle import ref_example as RE
from oracle import ref_numpy as R


class _OtherOrder(R.PolyCoefficientsConditional):
    """Same posterior; the force contraction summed in another order (data
    range in 7 chunks, reversed inside each chunk) and the mock data from the
    design-matrix product instead of Horner -- the kinds of difference a GPU
    kernel has."""

    def gradient(self, **variables):
        c = variables[self.variable_name]
        J = self.jacobi_matrix(c)
        r = (J.T.dot(c) - self.ys) * self.precision
        parts = [J[:, ch][:, ::-1].dot(r[ch][::-1])
                 for ch in np.array_split(np.arange(len(r)), 7)]
        return np.sum(parts, axis=0)


class _Injected(R.PolyCoefficientsConditional):
    """Force = the oracle's + s * eps * B with fixed random signs s."""

    def __init__(self, *a, **kw):
        self.eps = kw.pop('eps')
        self.rs = np.random.RandomState(kw.pop('seed'))
        R.PolyCoefficientsConditional.__init__(self, *a, **kw)

    def gradient(self, **variables):
        c = variables[self.variable_name]
        g = R.PolyCoefficientsConditional.gradient(self, **variables)
        J = self.jacobi_matrix(c)
        B = np.abs(J).dot(np.abs((R.polyval(self.xses, c) - self.ys) * self.precision))
        return g + self.rs.choice([-1.0, 1.0], size=len(c)) * self.eps * B


CASES = [(4, 20, 50, 0.02, 2.0), (7, 37, 9, 0.01, 1.5), (16, 128, 5, 2e-3, 1.5),
         (33, 2048, 5, 2e-4, 1.0), (33, 2048, 20, 2e-4, 1.0)]


def _setup(K, N, xlim, seed):
    rs = np.random.RandomState(seed)
    xs = np.linspace(-xlim, xlim, N)
    ys = R.polyval(xs, rs.standard_normal(K)) + rs.standard_normal(N) / np.sqrt(2.5)
    q0 = 0.3 * rs.standard_normal(K)
    p0 = rs.standard_normal(K)
    return xs, ys, q0, p0


def _run(cls, xs, ys, tau, K, q0, p0, dt, L, **kw):
    pdf = cls(xs, ys, tau, prior_means=np.zeros(K), prior_variances=np.ones(K) * 5,
              gamma_shape=1.0, gamma_rate=1.0, **kw)
    s = R.RefHMCSampler(pdf, q0.copy(), dt, L, variable_name='coefficients',
                        normal=lambda size: p0.copy(), uniform=lambda: 0.0)
    q = s.state.copy()
    p = p0.copy()
    q, p = s._leapfrog(q, p, dt, L)
    s2 = R.RefHMCSampler(pdf, q0.copy(), dt, L, variable_name='coefficients',
                         normal=lambda size: p0.copy(), uniform=lambda: 0.0)
    s2.sample()
    return q, p, s2.last_E_before, s2.last_E_after


@pytest.mark.parametrize('K,N,L,dt,xlim', CASES)
def test_other_summation_order_stays_inside_the_bound(K, N, L, dt, xlim):
    xs, ys, q0, p0 = _setup(K, N, xlim, K + N + L)
    tau = 2.5
    q_ref, p_ref, eb_ref, ea_ref = _run(R.PolyCoefficientsConditional, xs, ys, tau, K, q0, p0, dt, L)
    q_alt, p_alt, eb_alt, ea_alt = _run(_OtherOrder, xs, ys, tau, K, q0, p0, dt, L)
    pb = PB.PolyBound(xs, ys, K, np.zeros(K), np.ones(K) * 5)
    # the worst case of ANY summation order on the B scale is ~ (N + K) u: the
    # propagation must hold with that eps already (far below the 1e-10 bar)
    b = pb.transition(q0, p0, tau, dt, L, eps=4 * (N + K) * PB.U)
    assert np.all(np.abs(q_alt - q_ref) <= b['bq'] + 4 * PB.U * np.abs(q_ref))
    assert np.all(np.abs(p_alt - p_ref) <= b['bp'] + 4 * PB.U * np.abs(p_ref))
    assert eb_alt == eb_ref                                  # same state, same order
    assert abs(ea_alt - ea_ref) <= b['be_after']
    big = pb.transition(q0, p0, tau, dt, L)                   # EPS = 1e-10
    assert np.all(big['bq'] >= b['bq']) and big['be_after'] >= b['be_after']


@pytest.mark.parametrize('K,N,L,dt,xlim', CASES)
def test_injected_force_error_is_bounded_and_the_bound_is_not_vacuous(K, N, L, dt, xlim):
    xs, ys, q0, p0 = _setup(K, N, xlim, 3 * K + N)
    tau = 1.7
    q_ref, p_ref, _, ea_ref = _run(R.PolyCoefficientsConditional, xs, ys, tau, K, q0, p0, dt, L)
    pb = PB.PolyBound(xs, ys, K, np.zeros(K), np.ones(K) * 5)
    b = pb.transition(q0, p0, tau, dt, L)
    worst = 0.0
    for seed in range(5):
        q_inj, p_inj, _, ea_inj = _run(_Injected, xs, ys, tau, K, q0, p0, dt, L,
                                       eps=PB.EPS, seed=seed)
        dq, dp = np.abs(q_inj - q_ref), np.abs(p_inj - p_ref)
        # first-order bound; the second-order remainder is ~ EPS^2
        assert np.all(dq <= 1.001 * b['bq'] + 4 * PB.U * np.abs(q_ref))
        assert np.all(dp <= 1.001 * b['bp'] + 4 * PB.U * np.abs(p_ref))
        assert abs(ea_inj - ea_ref) <= 1.001 * b['be_after']
        worst = max(worst, np.max(dq / b['bq']))
    assert worst > 0.02            # random signs reach a few % of the worst case


def test_start_state_and_precision_differences_propagate():
    K, N, L, dt = 4, 20, 50, 0.02
    xs, ys = RE.example_data()
    rs = np.random.RandomState(0)
    q0, p0 = np.ones(K) + 0.05 * rs.standard_normal(K), rs.standard_normal(K)
    tau = 2.0
    pb = PB.PolyBound(xs, ys, K, np.zeros(K), np.ones(K) * 5)
    dq0 = 1e-9 * rs.standard_normal(K)
    dtau = 3e-10 * tau
    q_ref, p_ref, eb_ref, ea_ref = _run(R.PolyCoefficientsConditional, xs, ys, tau, K, q0, p0, dt, L)
    q_alt, p_alt, eb_alt, ea_alt = _run(R.PolyCoefficientsConditional, xs, ys, tau + dtau, K,
                                        q0 + dq0, p0, dt, L)
    b = pb.transition(q0, p0, tau, dt, L, bq0=np.abs(dq0), btau=dtau / tau, eps=0.0)
    assert np.all(np.abs(q_alt - q_ref) <= 1.001 * b['bq'] + 1e-15)
    assert np.all(np.abs(p_alt - p_ref) <= 1.001 * b['bp'] + 1e-15)
    assert abs(eb_alt - eb_ref) <= 1.001 * b['be_before']
    assert abs(ea_alt - ea_ref) <= 1.001 * b['be_after']
    # conjugate update: relative bound on the new precision
    bt = pb.gamma_update(q_ref, np.abs(q_alt - q_ref), 1.0)
    r0 = R.gamma_rate(xs, ys, q_ref, 1.0)
    r1 = R.gamma_rate(xs, ys, q_alt, 1.0)
    assert abs(r1 - r0) / r0 <= bt

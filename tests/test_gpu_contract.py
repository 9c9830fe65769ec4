"""GPU tests of the generic chain-rule contraction (``binf_jacobian_contract_f64``,
Likelihood._evaluate_gradient, binf/pdf/likelihoods.py:148-155: ``dfm.dot(emgrad)``)
and of the Posterior's term sums (``binf_sum_terms_f64``, posteriors.py:147-151,
173-187).

Bars: the contraction is held to 1e-10 of ``sum_n |J||r|`` against numpy (the
reference's BLAS order is not reproducible; exact on integer data, which also pins
the MFMA operand maps); the term sum is BIT-EXACT against Python's left-to-right
``+``.  A user forward model WITHOUT a fused kernel then runs a full sample() on
the library's own kernels and agrees with the numpy restatement
(oracle/ref_numpy.py:PolyCoefficientsConditional)."""
import numpy as np
import pytest
import torch

from binf_amd import _native
from binf_amd.example.likelihood import POLYVAL, ForwardModel, GaussianErrorModel
from binf_amd.example.priors import GammaPrior, GaussianPrior
from binf_amd.pdf.likelihoods import Likelihood, contract_jacobian
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers.hmc import HMCSampler
import poly_bounds as PB
from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu


def dev_t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)


SHAPES = [(1, 1, 1), (3, 5, 2), (16, 64, 16), (17, 65, 17), (33, 1000, 20), (4, 20, 100),
          (64, 63, 15), (65, 129, 33), (130, 300, 5), (32, 4096, 40), (7, 1, 9)]


@pytest.mark.parametrize('K,N,C', SHAPES)
@pytest.mark.parametrize('batched', [False, True])
def test_contraction_matches_numpy_within_the_bar(device, K, N, C, batched):
    rs = np.random.RandomState(K * 1000 + N + C)
    J = rs.standard_normal((C, K, N) if batched else (K, N))
    r = rs.standard_normal((C, N))
    got = _native.jacobian_contract(dev_t(J, device), dev_t(r, device)).cpu().numpy()
    assert got.shape == (C, K)
    for c in range(C):
        Jc = J[c] if batched else J
        want = Jc.dot(r[c])
        bar = 1e-10 * np.abs(Jc).dot(np.abs(r[c]))
        assert np.all(np.abs(got[c] - want) <= bar), (c, np.abs(got[c] - want).max(), bar.min())
        # ... and it uses a sliver of the allowance: any summation order is within N u
        assert np.all(np.abs(got[c] - want) <= (N + 2) * PB.U * np.abs(Jc).dot(np.abs(r[c])) + 1e-300)


@pytest.mark.parametrize('K,N,C', [(16, 64, 16), (33, 200, 19), (70, 130, 5)])
@pytest.mark.parametrize('batched', [False, True])
def test_contraction_is_exact_on_integer_data_with_asymmetric_operands(device, K, N, C, batched):
    """Small integers: every product and partial sum is exact, so any mix-up of rows,
    columns or chains (a wrong MFMA operand map) shows as an integer difference."""
    rs = np.random.RandomState(5)
    J = rs.randint(-9, 10, size=(C, K, N) if batched else (K, N)).astype(np.float64)
    J += (np.arange(K)[:, None] * 3 + np.arange(N)[None, :] * 7) % 11      # asymmetric
    r = rs.randint(-9, 10, size=(C, N)).astype(np.float64) + np.arange(C)[:, None]
    got = _native.jacobian_contract(dev_t(J, device), dev_t(r, device)).cpu().numpy()
    want = np.einsum('ckn,cn->ck', J, r) if batched else r.dot(J.T)
    assert np.array_equal(got, want)


def test_contraction_order_does_not_depend_on_the_batch(device):
    K, N, C = 33, 777, 50
    rs = np.random.RandomState(1)
    J, r = rs.standard_normal((K, N)), rs.standard_normal((C, N))
    Jb = rs.standard_normal((C, K, N))
    full = _native.jacobian_contract(dev_t(J, device), dev_t(r, device))
    fullb = _native.jacobian_contract(dev_t(Jb, device), dev_t(r, device))
    for lo, hi in ((0, 1), (3, 20), (17, 50), (49, 50)):
        part = _native.jacobian_contract(dev_t(J, device), dev_t(r[lo:hi], device))
        assert torch.equal(part, full[lo:hi])
        partb = _native.jacobian_contract(dev_t(Jb[lo:hi], device), dev_t(r[lo:hi], device))
        assert torch.equal(partb, fullb[lo:hi])
    # one chain, the reference's own shapes: [K x N] . [N] -> [K]
    one = contract_jacobian(dev_t(J, device), dev_t(r[7], device))
    assert one.shape == (K,) and torch.equal(one, full[7])


@pytest.mark.parametrize('K,N', [(5, 1024), (20, 1100), (33, 2049), (50, 1024), (64, 1500)])
def test_big_batches_take_32_chain_workgroups_with_the_same_bits(device, K, N):
    """From 8192 chains and 1024 data points up a shared Jacobian is contracted by workgroups of
    32 chains / 8 waves (csrc/jacobian.hip: half the reads of J through L2): a chain's sums are
    formed in the same order as in the 16-chain workgroups a small batch takes -- the first, a
    middle and the last (ragged) chains recomputed as small batches equal the big run's rows bit
    for bit -- and match numpy within the bar."""
    C = 8192 + 17
    rs = np.random.RandomState(K * 1000 + N)
    J, r = rs.standard_normal((K, N)), rs.standard_normal((C, N))
    tJ, tr = dev_t(J, device), dev_t(r, device)
    full = _native.jacobian_contract(tJ, tr)
    for lo, hi in ((0, 40), (4090, 4130), (C - 49, C), (C - 1, C)):
        part = _native.jacobian_contract(tJ, tr[lo:hi].contiguous())
        assert torch.equal(part, full[lo:hi])
    want = r[:64].dot(J.T)
    bound = 1e-10 * np.abs(r[:64]).dot(np.abs(J).T)
    assert np.all(np.abs(full[:64].cpu().numpy() - want) <= bound)


def test_contraction_argument_checks(device):
    J = torch.zeros((3, 5), dtype=torch.float64, device=device)
    with pytest.raises(ValueError):
        _native.jacobian_contract(J, torch.zeros((2, 6), dtype=torch.float64, device=device))
    with pytest.raises(ValueError):
        _native.jacobian_contract(torch.zeros((4, 3, 5), dtype=torch.float64, device=device),
                                  torch.zeros((2, 5), dtype=torch.float64, device=device))
    with pytest.raises(TypeError):            # no torch fallback for CPU tensors
        contract_jacobian(torch.zeros((3, 5), dtype=torch.float64), torch.zeros(5, dtype=torch.float64))
    # numpy values: the host mirror (the reference's own [252, 396] case is in test_host_mirror)
    assert np.array_equal(contract_jacobian(np.eye(2), np.array([3.0, 4.0])), [3.0, 4.0])
    empty = _native.jacobian_contract(torch.zeros((3, 0), dtype=torch.float64, device=device),
                                      torch.zeros((4, 0), dtype=torch.float64, device=device))
    assert empty.shape == (4, 3) and float(empty.abs().sum()) == 0.0


def test_sum_terms_is_the_left_to_right_sum_bitwise(device):
    rs = np.random.RandomState(2)
    C = 1000
    vecs = [rs.standard_normal(C) * 10.0 ** rs.randint(-8, 8) for _ in range(5)]
    terms = [dev_t(vecs[0], device), 0.1, dev_t(vecs[1], device), dev_t(vecs[2], device), -7.3,
             dev_t(vecs[3], device), dev_t(vecs[4], device)]
    got = _native.sum_terms(terms).cpu().numpy()
    want = vecs[0] + 0.1
    want = want + vecs[1]
    want = want + vecs[2]
    want = want + (-7.3)
    want = want + vecs[3]
    want = want + vecs[4]
    assert np.array_equal(got, want)
    # matrices (the Posterior's gradient sum) and the limits
    a, b = rs.standard_normal((7, 5)), rs.standard_normal((7, 5))
    assert np.array_equal(_native.sum_terms([dev_t(a, device), dev_t(b, device)]).cpu().numpy(), a + b)
    with pytest.raises(ValueError):
        _native.sum_terms([dev_t(a, device)] * 17)
    with pytest.raises(ValueError):
        _native.sum_terms([dev_t(a, device), dev_t(a[:3], device)])
    with pytest.raises(TypeError):
        _native.sum_terms([1.0, 2.0])


class PlainPolynomial(ForwardModel):
    """A user's forward model: the same polynomial, but a subclass that overrides the
    evaluation -- the library must NOT route it to the fused polynomial kernels (it
    cannot know what the override does) and runs the generic plug-in path instead:
    forward model, error-model gradient, jacobi matrix, contraction."""

    def _evaluate(self, coefficients):
        return ForwardModel._evaluate(self, coefficients)


def user_posterior(xs, ys, K, tau):
    lik = Likelihood('points', PlainPolynomial(xs, POLYVAL), GaussianErrorModel(ys))
    assert lik._native_pair() is None
    post = Posterior({lik.name: lik}, {'precision_prior': GammaPrior(1.0, 0.2),
                                       'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K) * 5)})
    return post.conditional_factory(precision=tau)


@pytest.mark.parametrize('K,N,C', [(4, 20, 6), (7, 300, 4)])
def test_user_forward_model_gradient_and_log_prob_vs_restatement(device, K, N, C):
    rs = np.random.RandomState(3)
    xs = np.linspace(-1.5, 1.5, N)
    ys = R.polyval(xs, rs.standard_normal(K)) + 0.3 * rs.standard_normal(N)
    theta = rs.standard_normal((C, K))
    tau = 2.0
    cond = user_posterior(xs, ys, K, tau)
    assert cond.native_hmc_spec('coefficients') is None
    g = cond.gradient(coefficients=dev_t(theta, device)).cpu().numpy()
    lp = cond.log_prob(coefficients=dev_t(theta, device)).cpu().numpy()
    for c in range(C):
        ref = R.PolyCoefficientsConditional(xs, ys, tau, np.zeros(K), np.ones(K) * 5, 1.0, 1.0)
        want = ref.gradient(coefficients=theta[c])
        J = ref.jacobi_matrix(theta[c])
        resid = np.abs((R.polyval(xs, theta[c]) - ys) * tau)
        assert np.all(np.abs(g[c] - want) <= 1e-10 * np.abs(J).dot(resid))
        assert abs(lp[c] - ref.log_prob(coefficients=theta[c])) <= 1e-12 * abs(lp[c])


def test_user_forward_model_full_sample_on_the_per_step_tier(device):
    """HMCSampler.sample() on that posterior: the generic tier end to end, accept
    flags and states against the numpy restatement within the propagated 1e-10 bound."""
    K, N, C, L, dt = 4, 20, 8, 10, 0.01
    rs = np.random.RandomState(4)
    xs = np.linspace(-2, 2, N)
    ys = R.polyval(xs, np.array([2.0, -4.0, 1.0, 1.5])) + rs.standard_normal(N) / np.sqrt(2.5)
    theta = np.ones((C, K)) + 0.1 * rs.standard_normal((C, K))
    p0, u = rs.standard_normal((C, K)), rs.uniform(size=C)
    tau = 2.5
    s = HMCSampler(user_posterior(xs, ys, K, tau), dev_t(theta, device), dt, L,
                   variable_name='coefficients', record_energies=True)
    out = s.sample(p0=dev_t(p0, device), u=dev_t(u, device)).cpu().numpy()
    pb = PB.PolyBound(xs, ys, K, np.zeros(K), np.ones(K) * 5)
    want = R.hmc_sample_batch(
        lambda c: R.PolyCoefficientsConditional(xs, ys, tau, np.zeros(K), np.ones(K) * 5, 1.0, 1.0),
        theta, p0, u, dt, L, variable_name='coefficients')
    assert np.array_equal(s.last_move_accepted.cpu().numpy().astype(np.uint8), want['accepted'])
    assert want['accepted'].any()
    eb, ea = s.last_e_before.cpu().numpy(), s.last_e_after.cpu().numpy()
    for c in range(C):
        b = pb.transition(theta[c], p0[c], tau, dt, L)
        assert np.all(np.abs(out[c] - want['q_out'][c]) <= b['bq'] + 4 * PB.U * np.abs(want['q_out'][c])), c
        assert abs(eb[c] - want['e_before'][c]) <= b['be_before'], c
        assert abs(ea[c] - want['e_after'][c]) <= b['be_after'], c


# ---------------------------------------------------------------------------
# round 4: the Posterior's sums and zero force on the product path, no torch arithmetic
# ---------------------------------------------------------------------------
def test_posterior_sum_chunks_beyond_16_terms_and_broadcasts_device_scalars(device):
    """_sum_in_order: more than 16 terms go through several launches of the term-sum kernel
    with the running sum carried over (the same left-to-right order, bit for bit); a 0-dim
    device tensor is a broadcast device scalar (never read back to the host); Python floats
    ride along.  What the kernel cannot take is refused, not added with torch."""
    from binf_amd.pdf.posteriors import _sum_in_order
    rs = np.random.RandomState(4)
    C = 777
    vecs = [rs.standard_normal(C) * 10.0 ** rs.randint(-6, 6) for _ in range(37)]
    terms, want = [], None
    for i, v in enumerate(vecs):
        if i % 9 == 4:
            t, w = float(v[0]), float(v[0])                    # a Python float
        elif i % 9 == 7:
            t, w = torch.tensor(v[1], dtype=torch.float64, device=device), v[1]   # 0-dim device tensor
            assert t.dim() == 0
        else:
            t, w = dev_t(v, device), v
        terms.append(t)
        want = w if want is None else want + w
    got = _sum_in_order(terms)
    assert got.shape == (C,) and np.array_equal(got.cpu().numpy(), want)
    # exactly 16 and 17 terms (one launch / two)
    for T in (16, 17, 31, 32):
        g = _sum_in_order([dev_t(v, device) for v in vecs[:T]]).cpu().numpy()
        w = vecs[0]
        for v in vecs[1:T]:
            w = w + v
        assert np.array_equal(g, w), T
    # only device scalars and floats: a 0-dim device tensor comes back
    s = _sum_in_order([torch.tensor(1.5, dtype=torch.float64, device=device), 2.25,
                       torch.tensor(-0.125, dtype=torch.float64, device=device)])
    assert s.dim() == 0 and s.is_cuda and float(s) == (1.5 + 2.25) + -0.125
    # [C x D] gradients
    a, b, c = (rs.standard_normal((5, 7)) for _ in range(3))
    assert np.array_equal(_sum_in_order([dev_t(a, device), dev_t(b, device), dev_t(c, device)]).cpu().numpy(),
                          (a + b) + c)
    # refused: mixed shapes, non-contiguous, host tensors, other dtypes
    with pytest.raises(ValueError):
        _sum_in_order([dev_t(a, device), dev_t(a[:3], device)])
    with pytest.raises(ValueError):
        _sum_in_order([dev_t(a, device), dev_t(np.zeros((7, 5)), device).t()])
    with pytest.raises(TypeError):
        _sum_in_order([dev_t(a, device), torch.zeros((5, 7), dtype=torch.float64)])
    with pytest.raises(TypeError):
        _sum_in_order([dev_t(a, device), torch.zeros((5, 7), dtype=torch.float32, device=device)])
    # the reference's own unit-test values (plain floats / numpy) still add on the host
    assert _sum_in_order([1.0, 2.0, 3.5]) == 6.5
    assert np.array_equal(_sum_in_order([np.ones(3), np.arange(3.0)]), np.ones(3) + np.arange(3.0))


def test_broadcast_term_inside_the_output_is_refused(device):
    v = dev_t(np.arange(8.0), device)
    terms = [v, v[3]]                                  # v[3] is a 0-dim view INTO v
    out = _native.sum_terms(terms).cpu().numpy()       # separate output: fine
    assert np.array_equal(out, np.arange(8.0) + 3.0)
    import ctypes
    ptrs = (ctypes.c_void_p * 2)(v.data_ptr(), v[3].data_ptr())
    flags = (ctypes.c_uint8 * 2)(0, 1)
    rc = _native.lib().binf_sum_terms_bcast_f64(ptrs, None, flags, 2, v.data_ptr(), 8,
                                                _native.stream_handle(device))
    assert rc == _native.E_ALIAS


def test_posterior_without_a_differentiable_component_has_the_references_empty_force(device):
    """binf/pdf/posteriors.py:177-180: the zero vector has one entry per element of the
    DIFFERENTIABLE variables passed.  The example's GaussianPrior registers `coefficients` as
    non-differentiable (quirk Q4), so a posterior made of priors only has none: the reference
    returns numpy.zeros(0), and its HMCSampler._leapfrog then fails to broadcast it
    (`p -= 0.5 * timestep * gradient(q)`, hmc.py:116: ValueError).  Batched: [C x 0] on the
    device and the same ValueError -- not a length-C host vector."""
    K, C = 5, 12
    post = Posterior({}, {'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K) * 2.0)})
    assert post.variables == {'coefficients'} and post.differentiable_variables == set()
    q = dev_t(np.random.RandomState(0).standard_normal((C, K)), device)
    g = post.gradient(coefficients=q)
    assert isinstance(g, torch.Tensor) and g.is_cuda and tuple(g.shape) == (C, 0)
    assert tuple(post.gradient(coefficients=q[0]).shape) == (0,)
    lp = post.log_prob(coefficients=q)                      # ... while the energy has the prior
    assert np.allclose(lp.cpu().numpy(), -0.5 * np.sum(q.cpu().numpy() ** 2 / 2.0, axis=1), rtol=1e-13)
    s = HMCSampler(post, q, 0.1, 3, variable_name='coefficients')
    with pytest.raises(ValueError, match='differentiable'):
        s.sample(p0=torch.zeros_like(q), u=torch.zeros(C, dtype=torch.float64, device=device))
    with pytest.raises(ValueError, match='differentiable'):
        s._leapfrog(q.clone(), torch.ones_like(q), 0.1, 3)
    # the helper's other branch: differentiable variables passed but nothing to add (reachable
    # only through a subclass): zeros of the gradient's shape, on the device
    from binf_amd.pdf.posteriors import _zero_force
    z = _zero_force([q, q[:, :2]])
    assert z.is_cuda and tuple(z.shape) == (C, K + 2) and float(z.abs().sum()) == 0.0
    assert tuple(_zero_force([q]).shape) == (C, K)

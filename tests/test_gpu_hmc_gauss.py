"""GPU parity tests of the fused Gaussian HMC kernel and the generic per-step
tier, through the C ABI, against the oracle (bit-exact) and the committed
golden vectors.  Floating-point bar: EXACT mode is bit-identical; FMA mode is
within 1e-10 relative (BASELINE.json north_star) with identical accept flags
on the golden set."""
import numpy as np
import pytest
import torch

from binf_amd import _native
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
from conftest import golden_files, load_golden
from oracle import c_oracle
from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu

REL_TOL_FMA = 1e-10


def dev_t(a, device, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return t if dtype is None else t.to(dtype)


def run_fused(device, q0, p0, u, dt, L, k=1.0, x0=0.0, dt_chain=None,
              adapt=False, mode=_native.MODE_EXACT, in_place=False):
    C, D = q0.shape
    tq0, tp0, tu = dev_t(q0, device), dev_t(p0, device), dev_t(u, device)
    q_out = tq0 if in_place else torch.empty_like(tq0)
    acc = torch.empty(C, dtype=torch.uint8, device=device)
    nacc = torch.zeros(C, dtype=torch.int64, device=device)
    eb = torch.empty(C, dtype=torch.float64, device=device)
    ea = torch.empty(C, dtype=torch.float64, device=device)
    tdt = dev_t(dt_chain, device) if dt_chain is not None else None
    _native.hmc_sample_gauss(tq0, tp0, tu, q_out, acc, nacc, eb, ea, dt, tdt, L,
                             k, x0, adapt, 1.05, 0.95, mode)
    torch.cuda.synchronize()
    return dict(q_out=q_out.cpu().numpy(), accepted=acc.cpu().numpy(),
                e_before=eb.cpu().numpy(), e_after=ea.cpu().numpy(),
                n_accepted=nacc.cpu().numpy(),
                timestep_out=None if tdt is None else tdt.cpu().numpy())


def assert_bitwise(got, want, keys=('q_out', 'accepted', 'e_before', 'e_after')):
    for k in keys:
        assert np.array_equal(got[k], want[k]), k


# --------------------------------------------------------------------------
# golden vectors
# --------------------------------------------------------------------------
def fused_covers(D):
    return D <= 8192 and _native.pairwise_tree_height(D) <= 6


@pytest.mark.parametrize('path', golden_files('gauss_'))
def test_sampler_reproduces_golden_vectors(device, path):
    """Through the HMCSampler surface (fused or generic tier, whichever the
    dispatch picks), every call of every golden set, attributes included."""
    g = load_golden(path)
    D, L = int(g['D']), int(g['L'])
    limit = int(g['adaption_limit'])
    ncalls, C, _ = g['p0'].shape
    s = HMCSampler(IsotropicGaussian(float(g['k']), float(g['x0'])),
                   dev_t(g['q0'], device), float(g['timestep']), L,
                   timestep_adaption_limit=limit, variable_name='x',
                   record_energies=True)
    for i in range(ncalls):
        out = s.sample(p0=dev_t(g['p0'][i], device), u=dev_t(g['u'][i], device))
        assert np.array_equal(out.cpu().numpy(), g['q_out'][i]), 'call %d' % i
        assert np.array_equal(s.last_move_accepted.cpu().numpy(),
                              g['accepted'][i].astype(bool))
        assert np.array_equal(s.last_e_before.cpu().numpy(), g['e_before'][i])
        assert np.array_equal(s.last_e_after.cpu().numpy(), g['e_after'][i])
        if limit:
            assert np.array_equal(s.timestep.cpu().numpy(), g['timestep_out'][i])
    assert np.array_equal(s.n_accepted.cpu().numpy(),
                          g['accepted'].sum(axis=0).astype(np.int64))


@pytest.mark.parametrize('path', golden_files('gauss_'))
def test_fused_kernel_reproduces_golden_vectors(device, path):
    g = load_golden(path)
    D, L = int(g['D']), int(g['L'])
    if not fused_covers(D):
        pytest.skip('D=%d: pairwise tree height > 3, generic tier (covered by '
                    'test_sampler_reproduces_golden_vectors)' % D)
    k, x0, dt0 = float(g['k']), float(g['x0']), float(g['timestep'])
    limit = int(g['adaption_limit'])
    ncalls, C, _ = g['p0'].shape
    q = g['q0'].copy()
    dt = np.full(C, dt0)
    for i in range(ncalls):
        adapt = (i + 1) < limit
        r = run_fused(device, q, g['p0'][i], g['u'][i], dt0, L, k, x0,
                      dt_chain=dt if limit else None, adapt=adapt)
        assert np.array_equal(r['q_out'], g['q_out'][i]), 'q_out call %d' % i
        assert np.array_equal(r['accepted'], g['accepted'][i])
        assert np.array_equal(r['e_before'], g['e_before'][i])
        assert np.array_equal(r['e_after'], g['e_after'][i])
        if limit:
            assert np.array_equal(r['timestep_out'], g['timestep_out'][i])
            dt = r['timestep_out']
        q = r['q_out']


@pytest.mark.parametrize('path', golden_files('gauss_'))
def test_fma_mode_within_tolerance_and_same_flags_on_golden(device, path):
    g = load_golden(path)
    if int(g['adaption_limit']):
        pytest.skip('adaption sets use per-call state; covered in exact mode')
    if not fused_covers(int(g['D'])):
        pytest.skip('generic tier')
    r = run_fused(device, g['q0'], g['p0'][0], g['u'][0], float(g['timestep']),
                  int(g['L']), float(g['k']), float(g['x0']),
                  mode=_native.MODE_FMA)
    assert np.array_equal(r['accepted'], g['accepted'][0])
    scale = np.abs(g['q_out'][0]).max()
    assert np.abs(r['q_out'] - g['q_out'][0]).max() <= REL_TOL_FMA * scale
    assert np.allclose(r['e_after'], g['e_after'][0], rtol=REL_TOL_FMA, atol=0)


# --------------------------------------------------------------------------
# random sweep against the C oracle
# --------------------------------------------------------------------------
SWEEP = [
    # D, C, L, k, x0, dt
    (1, 9, 3, 1.0, 0.0, 0.9), (2, 64, 2, 2.5, 0.3, 0.6), (5, 17, 4, 1.0, 0.0, 0.7),
    (8, 33, 5, 1.0, 0.0, 0.6), (9, 8, 3, 1.0, -0.4, 0.6), (16, 100, 7, 2.5, 0.0, 0.4),
    (33, 257, 20, 1.0, 0.0, 0.35), (64, 11, 3, 1.0, 0.0, 0.5),
    (100, 50, 10, 2.5, 0.3, 0.2), (127, 5, 2, 1.0, 0.0, 0.4),
    (128, 64, 20, 1.0, 0.0, 0.3), (129, 7, 3, 1.0, 0.0, 0.3),
    (200, 13, 6, 1.0, 0.1, 0.3), (256, 40, 20, 1.0, 0.0, 0.3),
    (258, 6, 2, 1.0, 0.0, 0.3), (260, 6, 2, 2.5, 0.3, 0.2), (300, 21, 9, 1.0, 0.0, 0.25),
    (512, 33, 20, 1.0, 0.0, 0.25), (520, 9, 2, 1.0, 0.0, 0.25),
    (768, 65, 20, 1.0, 0.0, 0.22), (776, 4, 3, 1.0, 0.0, 0.22),
    (900, 5, 2, 1.0, 0.0, 0.2), (920, 4, 2, 2.5, 0.3, 0.1),
    (1000, 7, 3, 1.0, 0.0, 0.2), (1024, 130, 20, 1.0, 0.0, 0.2),
    (1024, 3, 50, 2.5, 0.3, 0.1),
    # chains spanning 2 / 4 / 8 waves
    (1023, 5, 3, 1.0, 0.0, 0.2), (1025, 5, 3, 1.0, 0.0, 0.2), (1100, 3, 2, 2.5, 0.3, 0.1),
    (2000, 7, 4, 1.0, 0.0, 0.15), (2046, 2, 2, 1.0, 0.0, 0.15), (2048, 9, 20, 1.0, 0.0, 0.15),
    (3000, 3, 3, 1.0, -0.1, 0.1), (4096, 5, 5, 1.0, 0.0, 0.1), (5000, 2, 2, 2.5, 0.0, 0.05),
    (7400, 2, 2, 1.0, 0.0, 0.08), (8192, 3, 4, 1.0, 0.0, 0.08),
]


@pytest.mark.parametrize('D,C,L,k,x0,dt', SWEEP)
def test_fused_kernel_bitwise_vs_oracle(device, D, C, L, k, x0, dt):
    rs = np.random.RandomState(D * 7 + C)
    q0 = rs.standard_normal((C, D))
    p0 = rs.standard_normal((C, D))
    u = rs.uniform(size=C)
    want = c_oracle.hmc_sample_gauss(q0, p0, u, dt, L, k, x0, nthreads=4)
    got = run_fused(device, q0, p0, u, dt, L, k, x0)
    assert_bitwise(got, want)
    assert np.array_equal(got['n_accepted'], want['accepted'].astype(np.int64))
    assert 0 < want['accepted'].mean() or C < 8     # sweep has accepts ...


@pytest.mark.parametrize('D,C,L,k,x0,dt', SWEEP)
def test_fma_mode_bitwise_vs_the_fused_oracle(device, D, C, L, k, x0, dt):
    """FMA mode is not "EXACT within a tolerance" but an arithmetic of its own: each leapfrog
    update one correctly rounded fused multiply-add (``c_oracle.hmc_sample_gauss(fma=True)``,
    re-derived in rational arithmetic in tests/test_oracle.py).  Every kernel layout gives those
    bits -- states, energies, flags -- and stays within 1e-10 of EXACT mode."""
    rs = np.random.RandomState(1000 + D * 7 + C)
    q0, p0, u = rs.standard_normal((C, D)) + x0, rs.standard_normal((C, D)), rs.uniform(size=C)
    want = c_oracle.hmc_sample_gauss(q0, p0, u, dt, L, k=k, x0=x0, fma=True)
    assert_bitwise(run_fused(device, q0, p0, u, dt, L, k=k, x0=x0, mode=_native.MODE_FMA), want)
    exact = c_oracle.hmc_sample_gauss(q0, p0, u, dt, L, k=k, x0=x0)
    same = want['accepted'] == exact['accepted']
    assert np.abs(want['q_out'][same] - exact['q_out'][same]).max() <= REL_TOL_FMA * np.abs(exact['q_out']).max()


def test_fma_mode_bitwise_with_adaption_on_the_per_step_tier_and_for_long_chains(device):
    rs = np.random.RandomState(77)
    # per-chain step sizes + adaption, the persistent kernel
    C, D, L = 70, 768, 5
    q0, p0, u, dts = rs.standard_normal((C, D)), rs.standard_normal((C, D)), rs.uniform(size=C), rs.uniform(0.05, 0.4, size=C)
    want = c_oracle.hmc_sample_gauss(q0, p0, u, dts, L, adapt=True, fma=True)
    got = run_fused(device, q0, p0, u, 9.0, L, dt_chain=dts, adapt=True, mode=_native.MODE_FMA)
    assert_bitwise(got, want)
    assert np.array_equal(got['timestep_out'], want['timestep_out'])
    # several transitions per launch
    n, C, D, L = 4, 2100, 256, 3
    q0, p0, u = rs.standard_normal((C, D)), rs.standard_normal((n, C, D)), rs.uniform(size=(n, C))
    s = HMCSampler(IsotropicGaussian(2.5, 0.3), dev_t(q0, device), 0.15, L, variable_name='x', mode='fma')
    rec = s.sample_n(n, p0=dev_t(p0, device), u=dev_t(u, device)).cpu().numpy()
    q = q0
    for i in range(n):
        q = c_oracle.hmc_sample_gauss(q, p0[i], u[i], 0.15, L, k=2.5, x0=0.3, fma=True, nthreads=8)['q_out']
        assert np.array_equal(rec[i], q), i
    # the per-step tier (kick / drift kernels) and the long-chain kernels
    for D, C in ((300, 16), (9000, 3), (16384, 2)):
        q0, p0, u = rs.standard_normal((C, D)), rs.standard_normal((C, D)), rs.uniform(size=C)
        want = c_oracle.hmc_sample_gauss(q0, p0, u, 0.02, 4, k=2.5, x0=0.3, fma=True)
        for generic in (False, True):
            pdf = IsotropicGaussian(2.5, 0.3)
            if generic:
                pdf.native_hmc_spec = lambda name: None
            s = HMCSampler(pdf, dev_t(q0, device), 0.02, 4, variable_name='x', mode='fma', record_energies=True)
            out = s.sample(p0=dev_t(p0, device), u=dev_t(u, device)).cpu().numpy()
            assert np.array_equal(out, want['q_out']), (D, generic)
            assert np.array_equal(s.last_e_after.cpu().numpy(), want['e_after'])
            assert np.array_equal(s.last_move_accepted.cpu().numpy(), want['accepted'].astype(bool))


def test_sweep_contains_rejections(device):
    rs = np.random.RandomState(5)
    C, D, L, dt = 256, 1024, 20, 0.2
    q0 = rs.standard_normal((C, D))
    p0 = rs.standard_normal((C, D))
    u = rs.uniform(size=C)
    want = c_oracle.hmc_sample_gauss(q0, p0, u, dt, L, nthreads=4)
    got = run_fused(device, q0, p0, u, dt, L)
    assert 0.05 < want['accepted'].mean() < 0.95
    assert_bitwise(got, want)
    rej = want['accepted'] == 0
    assert np.array_equal(got['q_out'][rej], q0[rej])     # hmc.py:163-164


def test_per_chain_timestep_and_adaption(device):
    rs = np.random.RandomState(11)
    C, D, L = 70, 768, 5
    q0 = rs.standard_normal((C, D))
    p0 = rs.standard_normal((C, D))
    u = rs.uniform(size=C)
    dts = rs.uniform(0.05, 0.4, size=C)
    want = c_oracle.hmc_sample_gauss(q0, p0, u, dts, L, adapt=True)
    got = run_fused(device, q0, p0, u, 123.0, L, dt_chain=dts, adapt=True)
    assert_bitwise(got, want)
    assert np.array_equal(got['timestep_out'], want['timestep_out'])
    up = want['accepted'] == 1                      # quirk Q3: uprate on accept
    assert np.array_equal(want['timestep_out'][up], dts[up] * 1.05)
    assert np.array_equal(want['timestep_out'][~up], dts[~up] * 0.95)


@pytest.mark.parametrize('D,C', [(64, 300), (128, 130), (256, 70), (512, 40), (768, 2100), (1024, 2100)])
@pytest.mark.parametrize('k,x0,mode', [(1.0, 0.0, _native.MODE_EXACT), (2.5, 0.3, _native.MODE_EXACT),
                                       (1.0, 0.0, _native.MODE_FMA)])
def test_uniform_step_size_instantiation_same_bits(device, D, C, k, x0, mode):
    """Without per-chain step sizes and without adaption the launcher picks the kernel
    instantiation that keeps the step size in scalar registers (hmc_gauss.hip:
    gauss_uniform_dt; regular one-wave-per-chain shapes).  Same chains, bit for bit, as
    the per-lane instantiation (forced here by handing over a dt_chain of equal entries),
    one transition or several per launch -- and the oracle's (EXACT mode)."""
    rs = np.random.RandomState(D + C)
    L, dt, n, thin = 3, 0.17, 4, 2
    q0, p0, u = rs.standard_normal((C, D)) + x0, rs.standard_normal((n, C, D)), rs.uniform(size=(n, C))
    uni = run_fused(device, q0, p0[0], u[0], dt, L, k=k, x0=x0, mode=mode)
    per = run_fused(device, q0, p0[0], u[0], 99.0, L, k=k, x0=x0, mode=mode, dt_chain=np.full(C, dt))
    assert_bitwise(uni, per)
    assert 0 < uni['accepted'].sum()
    if mode == _native.MODE_EXACT:
        assert_bitwise(uni, c_oracle.hmc_sample_gauss(q0, p0[0], u[0], dt, L, k=k, x0=x0))

    def many(dt_chain, timestep):
        tq = dev_t(q0, device)
        out, rec = torch.empty_like(tq), torch.empty((n // thin, C, D), dtype=torch.float64, device=device)
        acc = torch.empty((n, C), dtype=torch.uint8, device=device)
        nacc = torch.zeros(C, dtype=torch.int64, device=device)
        eb, ea = (torch.empty((n, C), dtype=torch.float64, device=device) for _ in range(2))
        _native.hmc_sample_n_gauss(tq, dev_t(p0, device), dev_t(u, device), out, rec, acc, nacc, eb, ea,
                                   timestep, dt_chain, L, n, thin, k, x0, 0, 1.05, 0.95, mode)
        return [t.cpu() for t in (out, rec, acc, nacc, eb, ea)]

    a = many(None, dt)
    b = many(torch.full((C,), dt, dtype=torch.float64, device=device), 99.0)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_in_place_state_update(device):
    rs = np.random.RandomState(12)
    C, D, L, dt = 96, 1024, 20, 0.2
    q0 = rs.standard_normal((C, D))
    p0 = rs.standard_normal((C, D))
    u = rs.uniform(size=C)
    want = c_oracle.hmc_sample_gauss(q0, p0, u, dt, L)
    got = run_fused(device, q0, p0, u, dt, L, in_place=True)
    assert np.array_equal(got['q_out'], want['q_out'])


def test_nan_and_divergent_chains_are_rejected_like_numpy(device):
    C, D, L = 8, 1024, 20
    rs = np.random.RandomState(13)
    q0 = rs.standard_normal((C, D))
    p0 = rs.standard_normal((C, D))
    u = rs.uniform(size=C)
    q0[1, 5] = np.nan            # NaN energy -> u < exp(nan) is False
    p0[2] *= 1e160               # overflow to inf/nan energies
    with np.errstate(all='ignore'):
        want = c_oracle.hmc_sample_gauss(q0, p0, u, 0.2, L)
    got = run_fused(device, q0, p0, u, 0.2, L)
    assert np.array_equal(got['accepted'], want['accepted'])
    assert got['accepted'][1] == 0 and got['accepted'][2] == 0
    assert np.array_equal(got['q_out'], want['q_out'], equal_nan=True)
    assert np.array_equal(got['e_after'], want['e_after'], equal_nan=True)


def test_clip_bounds_huge_energy_drop(device):
    # |dE| beyond the csb clip: exp(709) finite -> accept for any u < 1
    C, D, L = 8, 64, 1
    q0 = np.full((C, D), 40.0)
    p0 = np.zeros((C, D))
    u = np.full(C, 0.999999)
    want = c_oracle.hmc_sample_gauss(q0, p0, u, 1.0, L)
    got = run_fused(device, q0, p0, u, 1.0, L)
    assert_bitwise(got, want)


def test_argument_errors_raise_reference_exception_types(device):
    t = torch.zeros((4, 9000), dtype=torch.float64, device=device)
    u = torch.zeros(4, dtype=torch.float64, device=device)
    acc = torch.zeros(4, dtype=torch.uint8, device=device)
    with pytest.raises(NotImplementedError):       # D > 8192: not covered
        _native.hmc_sample_gauss(t, t.clone(), u, torch.empty_like(t), acc, None,
                                 None, None, 0.1, None, 3, 1.0, 0.0, False,
                                 1.05, 0.95)
    t = torch.zeros((4, 64), dtype=torch.float64, device=device)
    with pytest.raises(ValueError):                # nsteps < 1
        _native.hmc_sample_gauss(t, t.clone(), u, torch.empty_like(t), acc, None,
                                 None, None, 0.1, None, 0, 1.0, 0.0, False,
                                 1.05, 0.95)
    with pytest.raises(ValueError):                # wrong buffer size
        _native.hmc_sample_gauss(t, t.clone(), u[:3].contiguous(),
                                 torch.empty_like(t), acc, None, None, None,
                                 0.1, None, 3, 1.0, 0.0, False, 1.05, 0.95)


# --------------------------------------------------------------------------
# full BASELINE size (C2): size-independent properties + sampled bit check
# --------------------------------------------------------------------------
def test_c2_full_size_properties(device):
    C, D, L, dt = 4096, 1024, 20, 0.05
    q0 = np.random.RandomState(1234).standard_normal((C, D))
    p0 = np.random.RandomState(1000).standard_normal((C, D))
    u = np.random.RandomState(2000).uniform(size=C)
    got = run_fused(device, q0, p0, u, dt, L)
    # (1) full oracle check is cheap enough in C with threads
    want = c_oracle.hmc_sample_gauss(q0, p0, u, dt, L, nthreads=8)
    assert_bitwise(got, want)
    # (2) leapfrog is time-reversible: integrating back from (q, -p) returns q0
    #     (checked through the energies: E_before of the reversed move equals
    #     E_after of the forward move up to rounding)
    acc = got['accepted'].astype(bool)
    assert acc.mean() > 0.9
    # (3) energy error of the integrator is O(dt^2): tiny at dt = 0.05
    assert np.abs(got['e_after'] - got['e_before']).max() < 0.5
    # (4) rejected chains keep their state bit-for-bit
    assert np.array_equal(got['q_out'][~acc], q0[~acc])
    # (5) chains are independent: a permuted batch gives the permuted result
    perm = np.random.RandomState(3).permutation(C)
    got_p = run_fused(device, q0[perm], p0[perm], u[perm], dt, L)
    assert np.array_equal(got_p['q_out'], got['q_out'][perm])


# --------------------------------------------------------------------------
# generic tier
# --------------------------------------------------------------------------
ROW_LENGTHS = [0, 1, 3, 7, 8, 9, 33, 127, 128, 129, 200, 258, 300, 768, 1000,
               1023, 1024, 1025, 2049, 4096, 5000, 7688, 7689, 7700, 8191, 8192, 8193,
               15892, 16383, 16384, 20000, 50000, 100003]
# 7689..8191: the ragged lengths whose pairwise tree is 7 levels deep (65 leaves);
# 15892 = 8192 + 7700: such a chunk after a full one


@pytest.mark.parametrize('D', ROW_LENGTHS)
def test_row_sum_is_np_sum_bitwise(device, D):
    rs = np.random.RandomState(D + 1)
    C = 5 if D > 5000 else 37
    x = rs.standard_normal((C, D)) * 10 ** rs.uniform(-2, 2, size=(C, 1))
    tx = dev_t(x, device)
    for op, shift, scale, f in [
            (_native.ROW_SUM, 0.0, 1.0, lambda r: np.sum(r)),
            (_native.ROW_SUMSQ, 0.0, 0.5, lambda r: 0.5 * np.sum(r ** 2)),
            (_native.ROW_SUMSQ_SHIFT, 0.3, -1.25,
             lambda r: -1.25 * np.sum((r - 0.3) ** 2))]:
        got = _native.row_sum(tx, op, shift, scale).cpu().numpy()
        want = np.array([f(x[c]) for c in range(C)], dtype=np.float64)
        assert np.array_equal(got, want), (D, op)


def test_row_sum_signed_zero(device):
    x = torch.full((3, 5), -0.0, dtype=torch.float64, device=device)
    got = _native.row_sum(x).cpu().numpy()
    assert np.array_equal(got, np.zeros(3)) and not np.signbit(got).any()


class TorchGaussian(object):
    """A user-style plug-in PDF written with torch ops only: exercises the
    generic tier's contract (log_prob -> [C], gradient -> [C x D])."""

    def __init__(self, k, x0):
        self.k, self.x0 = k, x0

    def log_prob(self, x):
        s = _native.row_sum(x if x.dim() == 2 else x.reshape(1, -1),
                            _native.ROW_SUMSQ_SHIFT, shift=self.x0)
        return (-0.5 * self.k) * s

    def gradient(self, x):
        return self.k * (x - self.x0)


@pytest.mark.parametrize('D,C,L,k,x0,dt', [(33, 20, 4, 2.5, 0.3, 0.3),
                                           (1024, 32, 20, 1.0, 0.0, 0.2),
                                           (3000, 6, 3, 1.0, 0.1, 0.1),
                                           (9000, 3, 2, 1.0, 0.0, 0.05),
                                           (20000, 3, 2, 1.0, 0.0, 0.02)])
def test_generic_tier_bitwise_vs_oracle(device, D, C, L, k, x0, dt):
    rs = np.random.RandomState(D + L)
    q0 = rs.standard_normal((C, D))
    p0 = rs.standard_normal((C, D))
    u = rs.uniform(size=C)
    want = c_oracle.hmc_sample_gauss(q0, p0, u, dt, L, k, x0, nthreads=4)
    for pdf in (TorchGaussian(k, x0), IsotropicGaussian(k, x0)):
        s = HMCSampler(pdf, dev_t(q0, device), dt, L, variable_name='x')
        if isinstance(pdf, IsotropicGaussian):
            # force the generic tier
            pdf.native_hmc_spec = lambda name: None
        out = s.sample(p0=dev_t(p0, device), u=dev_t(u, device))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want['q_out'])
        assert np.array_equal(s.last_move_accepted.cpu().numpy(),
                              want['accepted'].astype(bool))
        assert np.array_equal(s.last_e_before.cpu().numpy(), want['e_before'])
        assert np.array_equal(s.last_e_after.cpu().numpy(), want['e_after'])


class TorchDoubleWell(object):
    """A non-Gaussian user pdf: log p = -a sum (x^2 - 1)^2; the same expression
    order as ``NumpyDoubleWell`` so the two round alike."""

    def __init__(self, a):
        self.a = a

    def log_prob(self, x):
        w = x * x - 1.0
        return (-self.a) * _native.row_sum((w * w).reshape(-1, x.shape[-1]))

    def gradient(self, x):
        return (4.0 * self.a) * x * (x * x - 1.0)


class NumpyDoubleWell(object):
    def __init__(self, a):
        self.a = a

    def log_prob(self, x):
        w = x * x - 1.0
        return (-self.a) * np.sum(w * w)

    def gradient(self, x):
        return (4.0 * self.a) * x * (x * x - 1.0)


@pytest.mark.parametrize('D,C,L,dt,adapt', [(7, 40, 5, 0.09, False), (300, 16, 8, 0.03, True),
                                            (2500, 4, 3, 0.01, False)])
def test_generic_tier_non_gaussian_pdf_bitwise_vs_oracle(device, D, C, L, dt, adapt):
    """The per-step tier with a pdf the library knows nothing about, three
    sample() calls in a row (rejections and the step-size adaption included),
    against the numpy restatement of the reference's sampler chain by chain."""
    rs = np.random.RandomState(D)
    a = 1.5
    q0 = rs.standard_normal((C, D))
    s = HMCSampler(TorchDoubleWell(a), dev_t(q0, device), dt, L, variable_name='x',
                   timestep_adaption_limit=3 if adapt else 0)   # adapts on calls 0 and 1
    q, dts = q0, np.full(C, dt)
    for call in range(3):
        p0 = rs.standard_normal((C, D))
        u = rs.uniform(size=C)
        want = R.hmc_sample_batch(lambda c: NumpyDoubleWell(a), q, p0, u, dts, L,
                                  adapt=adapt and call < 2)
        out = s.sample(p0=dev_t(p0, device), u=dev_t(u, device))
        assert np.array_equal(out.cpu().numpy(), want['q_out'])
        assert np.array_equal(s.last_move_accepted.cpu().numpy(), want['accepted'].astype(bool))
        assert np.array_equal(s.last_e_before.cpu().numpy(), want['e_before'])
        assert np.array_equal(s.last_e_after.cpu().numpy(), want['e_after'])
        q, dts = want['q_out'], want['timestep_out']
        if adapt:
            assert np.array_equal(np.broadcast_to(s.timestep.cpu().numpy(), (C,)), dts)

def test_leapfrog_method_has_the_reference_signature(device):
    """HMCSampler._leapfrog(q, p, timestep, nsteps) integrates in place and
    returns (q, p) like hmc.py:92-125 -- one chain as a [D] array, or a batch
    with a per-chain step size; sample() goes through it, so a subclass that
    overrides it changes the sampler."""
    rs = np.random.RandomState(5)
    D, L, dt = 37, 7, 0.11
    q0, p0 = rs.standard_normal(D), rs.standard_normal(D)
    ref = R.RefHMCSampler(R.GaussianPDF(2.5, 0.3), q0.copy(), dt, L, variable_name='x')
    qw, pw = ref._leapfrog(q0.copy(), p0.copy(), dt, L)
    s = HMCSampler(TorchGaussian(2.5, 0.3), dev_t(q0, device), dt, L, variable_name='x')
    q, p = dev_t(q0, device), dev_t(p0, device)
    rq, rp = s._leapfrog(q, p, dt, L)
    assert rq is q and rp is p
    assert np.array_equal(q.cpu().numpy(), qw) and np.array_equal(p.cpu().numpy(), pw)

    C = 5
    Q, P = rs.standard_normal((C, D)), rs.standard_normal((C, D))
    dts = np.linspace(0.05, 0.2, C)
    q, p = dev_t(Q, device), dev_t(P, device)
    s._leapfrog(q, p, dev_t(dts, device), L)
    for c in range(C):
        qw, pw = ref._leapfrog(Q[c].copy(), P[c].copy(), dts[c], L)
        assert np.array_equal(q[c].cpu().numpy(), qw) and np.array_equal(p[c].cpu().numpy(), pw)

    class Frozen(HMCSampler):
        def _leapfrog(self, q, p, timestep, nsteps):
            return q, p
    f = Frozen(TorchGaussian(1.0, 0.0), dev_t(Q, device), dt, L, variable_name='x')
    out = f.sample()
    assert torch.equal(out, dev_t(Q, device)) and bool(f.last_move_accepted.all())


# --------------------------------------------------------------------------
# the sampler class (reference surface)
# --------------------------------------------------------------------------
def test_hmcsampler_matches_reference_restatement_over_several_calls(device):
    """One chain, global np.random stream, adaption on: the batched sampler
    consumes the stream and updates its attributes exactly like the
    restatement of the reference class."""
    D, L, dt = 33, 7, 0.45
    q0 = np.random.RandomState(99).standard_normal(D)
    np.random.seed(4242)
    ref = R.RefHMCSampler(R.GaussianPDF(2.5, 0.3), q0.copy(), dt, L,
                          timestep_adaption_limit=4, variable_name='x')
    ref_out = [ref.sample().copy() for _ in range(6)]
    ref_acc = ref.n_accepted

    np.random.seed(4242)
    s = HMCSampler(IsotropicGaussian(2.5, 0.3), dev_t(q0, device), dt, L,
                   timestep_adaption_limit=4, variable_name='x')
    assert s.acceptance_rate == 0.0 and s.last_move_accepted == 0
    outs = []
    for _ in range(6):
        outs.append(s.sample())
    torch.cuda.synchronize()
    for a, b in zip(outs, ref_out):
        assert a.shape == (D,)
        assert np.array_equal(a.cpu().numpy(), b)
    assert s.counter == 6 == ref.counter
    assert int(s.n_accepted.sum()) == ref_acc
    assert float(s.timestep[0]) == ref.timestep
    assert float(s.acceptance_rate[0]) == ref.acceptance_rate
    stats = s.last_draw_stats
    assert list(stats) == ['x'] and bool(stats['x'].accepted[0]) == bool(ref.last_move_accepted)
    assert np.array_equal(s.state.cpu().numpy(), ref.state)


def test_hmcsampler_quirks(device):
    q0 = torch.zeros((2, 8), dtype=torch.float64, device=device)
    s = HMCSampler(IsotropicGaussian(), q0, 0.1, 3)         # variable_name=None
    assert s.variable_name == 'HMC'                         # hmc.py:80
    assert list(s.last_draw_stats) == ['HMC']
    with pytest.raises(TypeError):                          # quirk Q1
        s.sample()
    s2 = HMCSampler(IsotropicGaussian(), 1.0, 0.1, 12, variable_name='x')
    with pytest.raises(AttributeError):                     # quirk Q2
        s2.sample()


def test_constructor_and_call_edge_cases_follow_the_reference(device):
    """nsteps < 1 integrates ONE step (hmc.py:118-123: the loop runs nsteps - 1 times, the last
    drift and half kick always); a [C] tensor as ``timestep`` is a per-chain step size; thinning
    beyond n records nothing (an empty record, not None); host arrays are refused by name."""
    rs = np.random.RandomState(21)
    C, D = 5, 33
    q0, p0, u = rs.standard_normal((C, D)), rs.standard_normal((C, D)), rs.uniform(size=C)
    outs = {}
    for L in (1, 0, -2):
        s = HMCSampler(IsotropicGaussian(), dev_t(q0, device), 0.4, L, variable_name='x', record_energies=True)
        outs[L] = (s.sample(p0=dev_t(p0, device), u=dev_t(u, device)).cpu().numpy(), s.last_e_after.cpu().numpy())
        assert s.nsteps == L and s.leapfrog_steps == 1
        for c in range(C):
            ref = R.RefHMCSampler(R.GaussianPDF(), q0[c].copy(), 0.4, L, variable_name='x',
                                  normal=lambda size, c=c: p0[c].copy(), uniform=lambda c=c: u[c])
            assert np.array_equal(ref.sample(), outs[L][0][c]) and ref.last_E_after == outs[L][1][c]
        # the per-step tier (a PDF without a fused kernel) takes the same single step
        plain = IsotropicGaussian()
        plain.native_hmc_spec = lambda name: None
        g = HMCSampler(plain, dev_t(q0, device), 0.4, L, variable_name='x')
        assert np.array_equal(g.sample(p0=dev_t(p0, device), u=dev_t(u, device)).cpu().numpy(), outs[L][0])
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[-2][0], outs[1][0])

    dts = rs.uniform(0.05, 0.5, size=C)
    s = HMCSampler(IsotropicGaussian(), dev_t(q0, device), dev_t(dts, device), 3, variable_name='x')
    got = s.sample(p0=dev_t(p0, device), u=dev_t(u, device)).cpu().numpy()
    assert np.array_equal(got, c_oracle.hmc_sample_gauss(q0, p0, u, dts, 3)['q_out'])
    assert torch.equal(s.timestep, dev_t(dts, device))

    s = HMCSampler(IsotropicGaussian(), dev_t(q0, device), 0.1, 2, variable_name='x')
    rec = s.sample_n(3, thin=5, p0=dev_t(rs.standard_normal((3, C, D)), device), u=dev_t(rs.uniform(size=(3, C)), device))
    assert rec.shape == (0, C, D) and s.counter == 3
    plain = IsotropicGaussian()
    plain.native_hmc_spec = lambda name: None
    assert HMCSampler(plain, dev_t(q0, device), 0.1, 2, variable_name='x').sample_n(2, thin=3).shape == (0, C, D)
    with pytest.raises(TypeError, match='sampler state is a numpy'):
        HMCSampler(IsotropicGaussian(), q0, 0.1, 2, variable_name='x').sample()
    with pytest.raises(ValueError, match='must live in GPU memory'):
        HMCSampler(IsotropicGaussian(), torch.from_numpy(q0), 0.1, 2, variable_name='x').sample()


def test_sampler_never_mutates_tensors_it_handed_out(device):
    q0 = torch.randn((16, 256), dtype=torch.float64, device=device)
    keep = q0.clone()
    s = HMCSampler(IsotropicGaussian(), q0, 0.2, 5, variable_name='x')
    a = s.sample()
    a_copy = a.clone()
    s.sample()
    s.sample()
    torch.cuda.synchronize()
    assert torch.equal(q0, keep)          # quirk Q9: caller's array untouched
    assert torch.equal(a, a_copy)


# --------------------------------------------------------------------------
# persistent kernel: n transitions per launch == n single launches, bitwise
# --------------------------------------------------------------------------
@pytest.mark.parametrize('D,C,L,k,x0,dt,n,thin,limit', [
    (1024, 40, 20, 1.0, 0.0, 0.2, 6, 1, 0),
    (1024, 9, 5, 2.5, 0.3, 0.12, 7, 2, 0),
    (768, 33, 20, 1.0, 0.0, 0.22, 5, 5, 0),
    (33, 100, 7, 2.5, 0.3, 0.45, 9, 3, 6),
    (4, 17, 3, 1.0, 0.0, 0.8, 8, 1, 100),
    (258, 6, 4, 1.0, 0.1, 0.3, 4, 1, 3),
    (200, 12, 10, 1.0, 0.0, 0.55, 3, 1, 0),
    (2048, 5, 6, 1.0, 0.0, 0.15, 5, 2, 0),
    (3000, 3, 3, 2.5, 0.1, 0.06, 4, 1, 3),
    (8192, 2, 3, 1.0, 0.0, 0.08, 3, 1, 0),
])
def test_sample_n_equals_n_single_launches(device, D, C, L, k, x0, dt, n, thin, limit):
    rs = np.random.RandomState(D + n)
    q0 = rs.standard_normal((C, D))
    p0 = rs.standard_normal((n, C, D))
    u = rs.uniform(size=(n, C))
    a = HMCSampler(IsotropicGaussian(k, x0), dev_t(q0, device), dt, L,
                   timestep_adaption_limit=limit, variable_name='x',
                   record_energies=True)
    singles, acc1, eb1, ea1 = [], [], [], []
    for i in range(n):
        singles.append(a.sample(p0=dev_t(p0[i], device), u=dev_t(u[i], device)).cpu().numpy())
        acc1.append(a.last_move_accepted.cpu().numpy())
        eb1.append(a.last_e_before.cpu().numpy())
        ea1.append(a.last_e_after.cpu().numpy())
    b = HMCSampler(IsotropicGaussian(k, x0), dev_t(q0, device), dt, L,
                   timestep_adaption_limit=limit, variable_name='x',
                   record_energies=True)
    rec = b.sample_n(n, thin=thin, p0=dev_t(p0, device), u=dev_t(u, device))
    torch.cuda.synchronize()
    assert rec.shape == (n // thin, C, D)
    for r in range(n // thin):
        assert np.array_equal(rec[r].cpu().numpy(), singles[(r + 1) * thin - 1]), r
    assert np.array_equal(b.state.cpu().numpy(), singles[-1])
    assert np.array_equal(b.accepted_history.cpu().numpy(), np.stack(acc1))
    assert np.array_equal(b.last_e_before.cpu().numpy(), np.stack(eb1))
    assert np.array_equal(b.last_e_after.cpu().numpy(), np.stack(ea1))
    assert np.array_equal(b.n_accepted.cpu().numpy(), a.n_accepted.cpu().numpy())
    assert b.counter == a.counter == n
    if limit:
        assert np.array_equal(b.timestep.cpu().numpy(), a.timestep.cpu().numpy())
    if (D, n) in ((1024, 6), (33, 9), (768, 5)):          # these sets mix accepts and rejections
        assert 0 < np.stack(acc1).mean() < 1


@pytest.mark.parametrize('D,C', [(1024, 1), (1024, 7), (1024, 1024), (1024, 1025), (1024, 2047),
                                 (1024, 2048), (1024, 2049), (768, 3), (768, 1500)])
@pytest.mark.parametrize('k,x0,mode', [(1.0, 0.0, 'exact'), (2.5, 0.3, 'exact'), (1.0, 0.0, 'fma')])
def test_few_chains_spread_over_several_waves_same_bits(device, D, C, k, x0, mode):
    """Up to 1024 / 2048 chains of D = 768 or 1024 run with 4 / 2 waves per
    chain (csrc/hmc_gauss_split.hip); above that, one wave per chain.  The
    results must not depend on the layout: several transitions with per-chain
    adapting step sizes against the C oracle (exact mode), and against the
    one-wave kernel run on a padded batch (both modes)."""
    n, L, dt = 3, 6, 0.21
    rs = np.random.RandomState(D + C)
    q0 = rs.standard_normal((C, D)) * 1.1 + x0
    p0 = rs.standard_normal((n, C, D))
    u = rs.uniform(size=(n, C))
    assert _native.lib().binf_pairwise_tree_height(D) == 3

    def run(qq, pp, uu):
        s = HMCSampler(IsotropicGaussian(k, x0), dev_t(qq, device), dt, L,
                       timestep_adaption_limit=3, variable_name='x', mode=mode,
                       record_energies=True)
        rec = s.sample_n(n, p0=dev_t(pp, device), u=dev_t(uu, device))
        torch.cuda.synchronize()
        return (rec.cpu().numpy(), s.accepted_history.cpu().numpy(),
                s.last_e_before.cpu().numpy(), s.last_e_after.cpu().numpy(),
                s.timestep.cpu().numpy(), s.n_accepted.cpu().numpy())
    got = run(q0, p0, u)
    # the same chains inside a batch of more than 2048: one wave per chain
    pad = 2100
    qp = np.concatenate([q0, rs.standard_normal((pad, D))])
    pp = np.concatenate([p0, rs.standard_normal((n, pad, D))], axis=1)
    up = np.concatenate([u, rs.uniform(size=(n, pad))], axis=1)
    ref = run(qp, pp, up)
    for g, r in zip(got, ref):
        assert np.array_equal(g, r[:, :C] if r.ndim > 1 and r.shape[0] == n else r[:C])
    assert 0 < got[1].mean() < 1 or C < 4
    if mode == 'exact':
        q = q0.copy()
        dtc = np.full(C, dt)
        for i in range(n):
            want = c_oracle.hmc_sample_gauss(q, p0[i], u[i], dtc, L, k, x0,
                                             adapt=(i < 2), nthreads=8)
            assert np.array_equal(got[0][i], want['q_out'])
            assert np.array_equal(got[1][i], want['accepted'].astype(bool))
            assert np.array_equal(got[2][i], want['e_before'])
            assert np.array_equal(got[3][i], want['e_after'])
            q, dtc = want['q_out'], want['timestep_out']


def test_sample_n_generic_pdf_falls_back_to_a_loop(device):
    C, D, n = 5, 9000, 3
    rs = np.random.RandomState(0)
    q0, p0, u = rs.standard_normal((C, D)), rs.standard_normal((n, C, D)), rs.uniform(size=(n, C))
    s = HMCSampler(IsotropicGaussian(), dev_t(q0, device), 0.1, 3, variable_name='x')
    rec = s.sample_n(n, p0=dev_t(p0, device), u=dev_t(u, device))
    want = q0
    for i in range(n):
        want = c_oracle.hmc_sample_gauss(want, p0[i], u[i], 0.1, 3)['q_out']
    assert np.array_equal(rec[-1].cpu().numpy(), want)


# --------------------------------------------------------------------------
# chains of any length: the chunked fused path (csrc/hmc_gauss_big.hip)
# --------------------------------------------------------------------------
@pytest.mark.parametrize('D,C,L,k,x0,dt,adapt', [
    (8193, 3, 2, 1.0, 0.0, 0.05, False),       # one full chunk + one element
    (9000, 3, 2, 1.0, 0.0, 0.05, False),
    (16384, 5, 3, 2.5, 0.3, 0.03, True),       # two full chunks
    (20000, 3, 2, 1.0, 0.0, 0.02, False),
    (7689, 4, 2, 1.0, 0.0, 0.05, True),        # <= 8192 but tree height 7
    (8191, 2, 2, 2.5, -0.1, 0.04, False),
    (8192 * 3 + 7, 2, 2, 1.0, 0.0, 0.02, False),   # a last chunk shorter than one accumulator row
    (40000, 70, 1, 1.0, 0.0, 0.02, True)])
def test_long_chains_fused_bitwise_vs_oracle(device, D, C, L, k, x0, dt, adapt):
    """binf_hmc_sample_gauss_big_f64 through HMCSampler (what IsotropicGaussian
    selects beyond the persistent kernel's reach) against the C restatement:
    states, flags, energies and adapted step sizes bit for bit -- np.sum's
    8192-element chunk rule included."""
    assert not fused_covers(D)
    rs = np.random.RandomState(D + L)
    q0 = rs.standard_normal((C, D))
    p0 = rs.standard_normal((2, C, D))
    u = rs.uniform(size=(2, C))
    if adapt:
        u[0, ::2] = 0.999999           # make sure both branches of the adaption occur
        p0[0, ::2] *= 3.0
    s = HMCSampler(IsotropicGaussian(k, x0), dev_t(q0, device), dt, L, variable_name='x',
                   timestep_adaption_limit=10 if adapt else 0, record_energies=True)
    dts = np.full(C, dt)
    q = q0
    for i in range(2):
        want = c_oracle.hmc_sample_gauss(q, p0[i], u[i], dts if adapt else dt, L, k, x0,
                                         nthreads=4)
        out = s.sample(p0=dev_t(p0[i], device), u=dev_t(u[i], device))
        assert np.array_equal(out.cpu().numpy(), want['q_out']), i
        assert np.array_equal(s.last_move_accepted.cpu().numpy(), want['accepted'].astype(bool))
        assert np.array_equal(s.last_e_before.cpu().numpy(), want['e_before'])
        assert np.array_equal(s.last_e_after.cpu().numpy(), want['e_after'])
        if adapt:
            dts = np.where(want['accepted'].astype(bool), dts * 1.05, dts * 0.95)
            assert np.array_equal(s.timestep.cpu().numpy(), dts)
        q = want['q_out']
    if adapt:
        acc = s.n_accepted.cpu().numpy()
        assert 0 < acc.sum() < 2 * C


def test_long_chains_sample_n_fma_and_errors(device):
    D, C, L = 10000, 4, 3
    rs = np.random.RandomState(1)
    q0, p0, u = rs.standard_normal((C, D)), rs.standard_normal((3, C, D)), rs.uniform(size=(3, C))
    # sample_n loops over the chunked kernel: same bits as single calls
    a = HMCSampler(IsotropicGaussian(), dev_t(q0, device), 0.02, L, variable_name='x')
    rec = a.sample_n(3, p0=dev_t(p0, device), u=dev_t(u, device))
    b = HMCSampler(IsotropicGaussian(), dev_t(q0, device), 0.02, L, variable_name='x')
    for i in range(3):
        assert torch.equal(b.sample(p0=dev_t(p0[i], device), u=dev_t(u[i], device)), rec[i])
    # FMA mode: within 1e-10, same flags
    f = HMCSampler(IsotropicGaussian(), dev_t(q0, device), 0.02, L, variable_name='x', mode='fma')
    x = f.sample(p0=dev_t(p0[0], device), u=dev_t(u[0], device))
    assert torch.equal(f.last_move_accepted, a.accepted_history[0])
    assert np.allclose(x.cpu().numpy(), rec[0].cpu().numpy(), rtol=1e-10, atol=1e-12)
    # C ABI: aliasing and a short workspace are refused
    tq, tp, tu = dev_t(q0, device), dev_t(p0[0], device), dev_t(u[0], device)
    acc = torch.empty(C, dtype=torch.uint8, device=device)
    lib = _native.lib()
    need = lib.binf_hmc_sample_gauss_big_workspace_bytes(C, D)
    assert need == C * 2 * 4 * 8
    ws = torch.empty(need // 8, dtype=torch.float64, device=device)
    st = _native.stream_handle(device)
    args = lambda qo, wsb: (tq.data_ptr(), tp.data_ptr(), tu.data_ptr(), qo, acc.data_ptr(), None,
                            None, None, 0.02, None, C, D, L, 1.0, 0.0, 0, 1.05, 0.95, 0,
                            ws.data_ptr(), wsb, st)
    assert lib.binf_hmc_sample_gauss_big_f64(*args(tq.data_ptr(), need)) == _native.E_ALIAS
    out = torch.empty_like(tq)
    assert lib.binf_hmc_sample_gauss_big_f64(*args(out.data_ptr(), need - 8)) == _native.E_ARG
    assert lib.binf_hmc_sample_gauss_big_f64(*args(out.data_ptr(), need)) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, rec[0])


@pytest.mark.parametrize('C,D', [(1, 33), (5, 1024), (3, 9000)])
def test_sample_n_consumes_the_generator_like_n_sample_calls(device, C, D):
    """Without supplied draws sample_n(n) takes them from the sampler's generator
    in the order n sample() calls would (hmc.py:146,151: normal, uniform, normal,
    uniform ...): the same global numpy stream position afterwards and the same
    states, for the persistent kernel, the long-chain path, one chain as a [D] array;
    likewise for the stand-alone device generator."""
    from binf_amd.samplers.rng import DeviceRNG
    q0 = np.random.RandomState(D).standard_normal((C, D))
    state = dev_t(q0[0] if C == 1 else q0, device)
    np.random.seed(77)
    a = HMCSampler(IsotropicGaussian(), state, 0.1 / np.sqrt(D / 33.0), 3, variable_name='x')
    rec = a.sample_n(3)
    after_a = np.random.uniform()
    np.random.seed(77)
    b = HMCSampler(IsotropicGaussian(), state, 0.1 / np.sqrt(D / 33.0), 3, variable_name='x')
    for i in range(3):
        assert torch.equal(b.sample(), rec[i]), i
    assert np.random.uniform() == after_a
    assert rec.shape == (3,) + tuple(state.shape)
    a = HMCSampler(IsotropicGaussian(), state, 0.1, 2, variable_name='x',
                   rng=DeviceRNG(5, device, fused=False))
    b = HMCSampler(IsotropicGaussian(), state, 0.1, 2, variable_name='x',
                   rng=DeviceRNG(5, device, fused=False))
    rec = a.sample_n(4, thin=2)
    xs = [b.sample() for _ in range(4)]
    assert torch.equal(rec[0], xs[1]) and torch.equal(rec[1], xs[3])
    assert a.rng.offset == b.rng.offset == 8


def test_private_helpers_of_the_reference_surface(device):
    """_copy_state (hmc.py:127-134) and _adapt_timestep (hmc.py:183-191) exist as
    methods for code that calls or overrides them."""
    q0 = dev_t(np.random.RandomState(0).standard_normal((6, 33)), device)
    s = HMCSampler(IsotropicGaussian(), q0, 0.3, 3, variable_name='x')
    c = s._copy_state(q0)
    assert torch.equal(c, q0) and c.data_ptr() != q0.data_ptr()
    with pytest.raises(ValueError):
        s._adapt_timestep()
    rs = np.random.RandomState(1)
    u = rs.uniform(size=6)
    u[::2] = 0.9999999
    p0 = rs.standard_normal((6, 33))
    p0[::2] *= 6.0                                   # rejected
    s.sample(p0=dev_t(p0, device), u=dev_t(u, device))
    acc = s.last_move_accepted.cpu().numpy()
    assert acc.any() and not acc.all()
    assert s.timestep == 0.3                         # adaption limit 0: sample() does not adapt
    s._adapt_timestep()
    want = np.where(acc, 0.3 * 1.05, 0.3 * 0.95)
    assert np.array_equal(s.timestep.cpu().numpy(), want)


@pytest.mark.parametrize('D,C,n,thin,adapt', [(8193, 3, 5, 2, False), (16384, 2, 4, 2, True),
                                              (20000, 2, 3, 1, False), (40000, 1, 4, 3, True),
                                              (12000, 5, 6, 4, False)])
def test_long_chains_sample_n_one_call_vs_oracle_and_single_calls(device, D, C, n, thin, adapt):
    """binf_hmc_sample_n_gauss_big_f64 (n long-chain transitions from one call, every
    recorded state written where it is kept) against the C restatement, transition
    by transition, and against n single sample() calls: states, flags, energies,
    adapted step sizes bit for bit; q_out ends up right whether the last transition
    is recorded or not."""
    assert not fused_covers(D)
    L, dt = 3, 0.02
    rs = np.random.RandomState(D + n)
    q0 = rs.standard_normal((C, D))
    p0 = rs.standard_normal((n, C, D))
    u = rs.uniform(size=(n, C))
    if adapt:
        u[0, ::2] = 0.999999
        p0[0, ::2] *= 3.0
    lim = 4 if adapt else 0
    a = HMCSampler(IsotropicGaussian(), dev_t(q0, device), dt, L, variable_name='x',
                   timestep_adaption_limit=lim, record_energies=True)
    rec = a.sample_n(n, thin=thin, p0=dev_t(p0, device), u=dev_t(u, device))
    b = HMCSampler(IsotropicGaussian(), dev_t(q0, device), dt, L, variable_name='x',
                   timestep_adaption_limit=lim, record_energies=True)
    dts = np.full(C, dt)
    q = q0
    for i in range(n):
        want = c_oracle.hmc_sample_gauss(q, p0[i], u[i], dts if adapt else dt, L, nthreads=4)
        x = b.sample(p0=dev_t(p0[i], device), u=dev_t(u[i], device))
        assert np.array_equal(x.cpu().numpy(), want['q_out']), i
        if (i + 1) % thin == 0:
            assert torch.equal(rec[(i + 1) // thin - 1], x), i
        assert np.array_equal(a.accepted_history[i].cpu().numpy(), want['accepted'].astype(bool)), i
        assert np.array_equal(a.last_e_before[i].cpu().numpy(), want['e_before']), i
        assert np.array_equal(a.last_e_after[i].cpu().numpy(), want['e_after']), i
        if adapt and i + 1 < lim:
            dts = np.where(want['accepted'].astype(bool), dts * 1.05, dts * 0.95)
        q = want['q_out']
    assert rec.shape == (n // thin, C, D)
    assert torch.equal(a.state, b.state) and a.counter == b.counter == n
    assert torch.equal(a.n_accepted, b.n_accepted)
    if adapt:
        assert np.array_equal(a.timestep.cpu().numpy(), dts) and torch.equal(a.timestep, b.timestep)
    # nothing recorded: the state still arrives
    c = HMCSampler(IsotropicGaussian(), dev_t(q0, device), dt, L, variable_name='x',
                   timestep_adaption_limit=lim)
    assert c.sample_n(n, p0=dev_t(p0, device), u=dev_t(u, device), record=False) is None
    assert torch.equal(c.state, b.state)
    # into a caller's record buffer
    buf = torch.empty((n // thin, C, D), dtype=torch.float64, device=device)
    d = HMCSampler(IsotropicGaussian(), dev_t(q0, device), dt, L, variable_name='x',
                   timestep_adaption_limit=lim)
    r2 = d.sample_n(n, thin=thin, p0=dev_t(p0, device), u=dev_t(u, device), out=buf)
    assert r2.data_ptr() == buf.data_ptr() and torch.equal(buf, rec)


@pytest.mark.parametrize('fused', [True, False])
def test_long_chains_sample_n_draws_what_n_sample_calls_draw(device, fused):
    """With a device generator the one-call loop takes the stream positions n
    sample() calls take: lane streams in the kernels (fused) or the stand-alone
    Philox fills a block of transitions at a time -- identical chains either way."""
    D, C, n, L = 9000, 3, 5, 2
    q0 = np.random.RandomState(3).standard_normal((C, D))
    a = HMCSampler(IsotropicGaussian(2.0, 0.1), dev_t(q0, device), 0.01, L, variable_name='x',
                   rng=DeviceRNG(11, device, fused=fused), record_energies=True)
    b = HMCSampler(IsotropicGaussian(2.0, 0.1), dev_t(q0, device), 0.01, L, variable_name='x',
                   rng=DeviceRNG(11, device, fused=fused), record_energies=True)
    rec = a.sample_n(n, thin=2)
    xs = [b.sample().clone() for _ in range(n)]
    assert torch.equal(rec[0], xs[1]) and torch.equal(rec[1], xs[3])
    assert torch.equal(a.state, xs[-1]) and a.rng.offset == b.rng.offset
    assert torch.equal(a.last_e_after[-1], b.last_e_after)
    assert bool(a.accepted_history.any())


@pytest.mark.parametrize('C,D,L', [(64, 1024, 5), (4096, 1024, 3), (700, 1024, 4), (3, 9000, 2), (40, 300, 6)])
def test_sample_n_draws_what_n_sample_calls_draw(device, C, D, L):
    """A seed identifies the chains whatever the call shape: with a DeviceRNG whose draws are
    the lane streams of the fused kernels, transition i of sample_n(n) takes the stream
    position the i-th sample() call would take -- on the persistent kernel (in-kernel draws
    for large batches, the same draws written out first for small ones) and on the long-chain
    path (D = 9000) alike.  States, flags and energies equal bit for bit, and the generator
    ends at the same position."""
    n = 5
    rs = np.random.RandomState(C + D)
    q0 = rs.standard_normal((C, D))
    runs = []
    for shape in ('single', 'n', 'mixed'):
        rng = DeviceRNG(17, device)
        s = HMCSampler(IsotropicGaussian(1.0, 0.0), dev_t(q0, device), 0.15, L, variable_name='x',
                       rng=rng, record_energies=True)
        states, flags, ea = [], [], []

        def one():
            states.append(s.sample().clone())
            flags.append(s.last_move_accepted.clone())
            ea.append(s.last_e_after.reshape(-1).clone())

        def many(m):
            rec = s.sample_n(m)
            states.extend(rec[i].clone() for i in range(m))
            flags.extend(s.accepted_history[i].clone() for i in range(m))
            ea.extend(s.last_e_after.reshape(m, -1)[i].clone() for i in range(m))
        if shape == 'single':
            for _ in range(n):
                one()
        elif shape == 'n':
            many(n)
        else:
            one()
            many(3)
            one()
        runs.append((torch.stack(states), torch.stack(flags), torch.stack(ea), rng.offset))
    for other in runs[1:]:
        assert torch.equal(runs[0][0], other[0])
        assert torch.equal(runs[0][1], other[1])
        assert torch.equal(runs[0][2], other[2])
        assert runs[0][3] == other[3] == n
    assert not torch.equal(runs[0][0][0], runs[0][0][1])

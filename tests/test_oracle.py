"""CPU tests of the oracle itself (test infrastructure): the restatements agree
with numpy, with each other and with the committed golden vectors."""
import numpy as np
import pytest

from conftest import golden_files, load_golden
from oracle import c_oracle
from oracle import ref_numpy as R

LENGTHS = list(range(0, 300)) + [511, 512, 768, 920, 1000, 1023, 1024, 2049,
                                 4096, 5000, 7689, 7700, 8191, 15892, 16384, 100003]


def test_pairwise_restatement_matches_np_sum_bitwise():
    rs = np.random.RandomState(0)
    for n in LENGTHS:
        a = rs.standard_normal(n) * 10 ** rs.uniform(-3, 3)
        want = float(np.sum(a))
        if n <= 5000:
            assert R.np_sum_py(a) == want, n
        assert c_oracle.np_sum(a) == want, n


def test_pairwise_signed_zero():
    for n in (0, 1, 3, 8, 9, 200):
        a = np.full(n, -0.0)
        got = c_oracle.np_sum(a)
        want = float(np.sum(a))
        assert got == want and np.signbit(got) == np.signbit(want)


def test_pairwise_leaves_cover_the_vector():
    for n in (1, 7, 8, 128, 129, 258, 1023, 1024, 5000):
        leaves, _ = R.pairwise_leaves(n)
        pos = 0
        for off, ln in leaves:
            assert off == pos and 0 < ln <= 128
            pos += ln
        assert pos == n


def test_polyval_restatement_is_numpys():
    rs = np.random.RandomState(3)
    x = np.linspace(-2, 2, 50)
    for K in (1, 2, 4, 33):
        c = rs.standard_normal(K)
        assert np.array_equal(R.polyval(x, c),
                              np.polynomial.polynomial.polyval(x, c))


def test_clipped_exp_bounds():
    assert R.exp(-1e6) == np.exp(-308.0)
    assert R.exp(1e6) == np.exp(709.0)
    assert R.exp(0.5) == np.exp(0.5)
    assert np.isnan(R.exp(np.nan))


@pytest.mark.parametrize('D,L,k,x0,dt', [
    (1, 1, 1.0, 0.0, 0.5), (4, 3, 1.0, 0.0, 0.3), (7, 2, 2.5, 0.3, 0.9),
    (8, 5, 1.0, 0.0, 0.5), (33, 20, 2.5, 0.3, 0.35), (129, 4, 1.0, -1.0, 0.3),
    (300, 10, 1.0, 0.0, 0.25), (768, 20, 1.0, 0.0, 0.22),
    (1023, 3, 2.5, 0.3, 0.1), (1024, 20, 1.0, 0.0, 0.2), (2500, 2, 1.0, 0.0, 0.1),
    # longer than numpy's 8192-element reduction buffer: chunk sums chained
    (7689, 2, 1.0, 0.0, 0.05), (8193, 2, 1.0, 0.0, 0.05), (9000, 3, 2.5, 0.3, 0.03),
    (16389, 2, 1.0, 0.0, 0.03), (40000, 1, 1.0, -0.1, 0.02)])
def test_c_oracle_equals_numpy_restatement(D, L, k, x0, dt):
    rs = np.random.RandomState(D * 31 + L)
    C = 5
    q0 = rs.standard_normal((C, D))
    p0 = rs.standard_normal((C, D))
    u = rs.uniform(size=C)
    dts = dt * rs.uniform(0.8, 1.2, size=C)
    a = R.hmc_sample_batch(lambda c: R.GaussianPDF(k, x0), q0, p0, u, dts, L,
                           adapt=True)
    b = c_oracle.hmc_sample_gauss(q0, p0, u, dts, L, k, x0, adapt=True,
                                  nthreads=2)
    for key in a:
        assert np.array_equal(a[key], b[key]), key


def test_c_oracle_fma_mode_is_one_correctly_rounded_fused_multiply_add_per_update():
    """The checker of the package's FMA mode (``c_oracle.hmc_sample_gauss(fma=True)``): each leapfrog
    update ``p -= dt * g`` / ``q += p * dt`` as ONE fused multiply-add, everything else rounded as in
    EXACT mode.  Re-derived here in exact rational arithmetic (``float(Fraction)`` rounds correctly),
    sharing no code with the C file: same bits."""
    from fractions import Fraction as F

    def fma(a, b, c):
        return float(F(a) * F(b) + F(c))

    rs = np.random.RandomState(12)
    C, D, L, k, x0, dt = 3, 9, 4, 2.5, 0.3, 0.37
    q0, p0 = rs.standard_normal((C, D)), rs.standard_normal((C, D))
    got = c_oracle.hmc_sample_gauss(q0, p0, np.zeros(C), dt, L, k=k, x0=x0, fma=True)   # u = 0: accepted
    plain = c_oracle.hmc_sample_gauss(q0, p0, np.zeros(C), dt, L, k=k, x0=x0)
    assert got['accepted'].all() and not np.array_equal(got['q_out'], plain['q_out'])
    assert np.abs(got['q_out'] - plain['q_out']).max() < 1e-14
    for c in range(C):
        q, p = [float(v) for v in q0[c]], [float(v) for v in p0[c]]
        grad = lambda x: k * (x - x0)
        h = 0.5 * dt
        p = [fma(-h, grad(x), m) for x, m in zip(q, p)]
        for _ in range(L - 1):
            q = [fma(m, dt, x) for x, m in zip(q, p)]
            p = [fma(-dt, grad(x), m) for x, m in zip(q, p)]
        q = [fma(m, dt, x) for x, m in zip(q, p)]
        p = [fma(-h, grad(x), m) for x, m in zip(q, p)]
        assert np.array_equal(np.array(q), got['q_out'][c])
        want_e = 0.5 * k * np.sum((np.array(q) - x0) ** 2) + 0.5 * np.sum(np.array(p) ** 2)
        assert got['e_after'][c] == want_e


def test_adaption_multiplies_uprate_on_accept():
    # quirk Q3: reference hmc.py:188-191 (docstring says the opposite)
    s = R.RefHMCSampler(R.GaussianPDF(), np.zeros(4), 0.1, 1,
                        timestep_adaption_limit=10, variable_name='x',
                        normal=lambda size: np.zeros(size), uniform=lambda: 0.0)
    s.sample()
    assert s.last_move_accepted and s.timestep == 0.1 * 1.05


def test_adaption_limit_checked_after_increment():
    s = R.RefHMCSampler(R.GaussianPDF(), np.zeros(4), 0.1, 1,
                        timestep_adaption_limit=2, variable_name='x',
                        normal=lambda size: np.zeros(size), uniform=lambda: 0.0)
    s.sample()              # counter 1 < 2 -> adapts
    t1 = s.timestep
    s.sample()              # counter 2 == limit -> no adaption
    assert t1 == 0.1 * 1.05 and s.timestep == t1


@pytest.mark.parametrize('path', golden_files('gauss_'))
def test_golden_vectors_are_reproduced_by_both_restatements(path):
    g = load_golden(path)
    assert 'parity unpinned' in str(g['provenance'])
    D, L = int(g['D']), int(g['L'])
    k, x0, dt0 = float(g['k']), float(g['x0']), float(g['timestep'])
    limit = int(g['adaption_limit'])
    ncalls, C, _ = g['p0'].shape
    q = g['q0'].copy()
    dt = np.full(C, dt0)
    for i in range(ncalls):
        adapt = (i + 1) < limit
        r = c_oracle.hmc_sample_gauss(q, g['p0'][i], g['u'][i], dt, L, k, x0,
                                      adapt=adapt)
        assert np.array_equal(r['q_out'], g['q_out'][i])
        assert np.array_equal(r['accepted'], g['accepted'][i])
        assert np.array_equal(r['e_before'], g['e_before'][i])
        assert np.array_equal(r['e_after'], g['e_after'][i])
        assert np.array_equal(r['timestep_out'], g['timestep_out'][i])
        q, dt = r['q_out'], r['timestep_out']


def test_golden_rng_stream_order():
    """p0 / u in the fixtures are the global-stream draws in the reference's
    order: normal(size=D) then uniform() per sample (hmc.py:146,151)."""
    g = load_golden(golden_files('gauss_d33_l20')[0])
    seed = int(g['seed'])
    for c in range(g['p0'].shape[1]):
        np.random.seed(seed + c)
        for i in range(g['p0'].shape[0]):
            assert np.array_equal(np.random.normal(size=int(g['D'])), g['p0'][i, c])
            assert np.random.uniform() == g['u'][i, c]


@pytest.mark.parametrize('name', ['poly_c1_example', 'poly_k7_n37', 'poly_k16_n128',
                                  'poly_k33_n16384'])
def test_polynomial_gibbs_golden_is_what_the_restatement_gives(name):
    """The committed Gibbs-within-HMC fixtures (BASELINE C1 / C4 at CPU size)
    are the restatement's outputs for the reference's own stream consumption."""
    from oracle import gen_golden as G
    g = load_golden(golden_files(name)[0])
    spec = [s for s in G.POLY_SETS if s[0] == name][0]
    r = G.run_poly_set(*spec[1:])
    for k in ('xs', 'ys', 'p0', 'u', 'gamma', 'coefficients', 'precision', 'accepted',
              'e_before', 'e_after'):
        assert np.array_equal(r[k], g[k]), k
    if name == 'poly_c1_example':
        # example_script.py:17-26 with np.random.seed(0)
        np.random.seed(0)
        xs = np.linspace(-2, 2, 20)
        ys = np.random.normal(loc=np.polynomial.polynomial.polyval(xs, [2.0, -4.0, 1.0, 1.5]),
                              scale=1.0 / np.sqrt(2.5))
        assert np.array_equal(g['xs'], xs) and np.array_equal(g['ys'], ys)
        assert 0 < g['accepted'].mean() < 1


@pytest.mark.parametrize('name', ['dist_n12', 'dist_n40', 'dist_n100', 'dist_n256', 'dist_n300'])
def test_distance_golden_is_what_the_restatement_gives(name):
    from oracle import gen_golden as G
    g = load_golden(golden_files(name)[0])
    spec = [s for s in G.DIST_SETS if s[0] == name][0]
    r = G.run_dist_set(*spec[1:])
    for k in r:
        assert np.array_equal(r[k], g[k]), k


def test_c_oracle_under_asan_and_ubsan(tmp_path):
    """The checker itself under the sanitizers that ARE available here (CPU build; GPU ASan is not
    offered on this pool): oracle_c.c + oracle/sanitize_main.c built with -fsanitize=address,undefined
    and run over ragged lengths around the pairwise-sum leaf / chunk boundaries, exact-size heap
    buffers, one and two OpenMP threads."""
    import os
    import shutil
    import subprocess
    from conftest import ROOT
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    exe = str(tmp_path / 'oracle_san')
    cmd = [gcc, '-O1', '-g', '-fno-omit-frame-pointer', '-ffp-contract=off', '-fopenmp',
           '-fsanitize=address,undefined', '-fno-sanitize-recover=all',
           os.path.join(ROOT, 'oracle', 'oracle_c.c'), os.path.join(ROOT, 'oracle', 'sanitize_main.c'),
           '-o', exe, '-lm']
    b = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if b.returncode != 0 and b'sanitize' in b.stderr.lower():
        pytest.skip('this gcc has no sanitizer runtime: ' + b.stderr.decode()[-200:])
    assert b.returncode == 0, b.stderr.decode()[-2000:]
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', OMP_NUM_THREADS='2'))
    assert r.returncode == 0, (r.stdout.decode()[-500:], r.stderr.decode()[-3000:])
    assert b'sanitized oracle run ok' in r.stdout

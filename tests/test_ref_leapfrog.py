"""The oracle's integrator against outputs of THE REFERENCE'S OWN ``_leapfrog``.

``tests/golden/ref_leapfrog_*.npz`` were produced in the build container by
``oracle/gen_ref_leapfrog.py``: the reference's ``HMCSampler`` class compiled from
``binf/samplers/hmc.py`` itself with its one csb import statement dropped (csb is
absent; nothing was substituted for it), ``__init__`` and ``_leapfrog``
(``hmc.py:17-62,92-125``) called with duck-typed numpy PDFs.  This pins SURVEY 8(a)
row a2 -- the kick-drift-kick sequence, its roundings, the in-place update -- by
reference output; ``sample()``'s accept test and the csb ``exp`` clip stay "parity
unpinned" (nothing here touches them).

CPU: the numpy restatement (``RefHMCSampler._leapfrog``) and the C restatement
(``oracle_c.c``) reproduce those bits.  The HIP kernels are held to the same files
in ``tests/test_gpu_ref_leapfrog.py``."""
import numpy as np
import pytest

from conftest import golden_files, load_golden
from oracle import c_oracle
from oracle import ref_distance
from oracle import ref_numpy as R

FILES = golden_files('ref_leapfrog_')


def test_fixture_set_is_complete_and_says_where_it_comes_from():
    kinds = {}
    for f in FILES:
        g = load_golden(f)
        kinds.setdefault(str(g['kind']), []).append(f)
        prov = str(g['provenance'])
        assert 'REFERENCE' in prov and 'hmc.py:92-125' in prov and 'csb' in prov
        assert g['q_out'].shape == g['q0'].shape == g['p_out'].shape == g['p0'].shape
        assert not np.array_equal(g['q_out'], g['q0'])
    assert len(kinds['gauss']) >= 9 and len(kinds['poly']) == 2 and len(kinds['dist']) == 3


def _pdf(g):
    kind = str(g['kind'])
    if kind == 'gauss':
        return R.GaussianPDF(float(g['k']), float(g['x0']), variable_name='x'), 'x'
    if kind == 'poly':
        K = g['q0'].shape[1]
        return R.PolyCoefficientsConditional(g['xs'], g['ys'], float(g['precision']), np.zeros(K),
                                             np.ones(K), 1.0, 0.2), 'coefficients'
    return ref_distance.DistancePosterior(g['ys'], float(g['precision']), int(g['n_beads']),
                                          prior_k=float(g['prior_k'])), 'coordinates'


@pytest.mark.parametrize('path', FILES, ids=lambda p: p.split('ref_leapfrog_')[-1][:-4])
def test_numpy_restatement_reproduces_the_reference_integrator_bitwise(path):
    g = load_golden(path)
    pdf, name = _pdf(g)
    dts = np.broadcast_to(np.asarray(g['timestep'], dtype=np.float64), (g['q0'].shape[0],))
    for c in range(g['q0'].shape[0]):
        s = R.RefHMCSampler(pdf, g['q0'][c].copy(), float(dts[c]), int(g['nsteps']), variable_name=name)
        q, p = g['q0'][c].copy(), g['p0'][c].copy()
        s._leapfrog(q, p, s.timestep, s.nsteps)
        assert np.array_equal(q, g['q_out'][c]), (path, c)
        assert np.array_equal(p, g['p_out'][c]), (path, c)


@pytest.mark.parametrize('path', [f for f in FILES if 'gauss' in f],
                         ids=lambda p: p.split('ref_leapfrog_')[-1][:-4])
def test_c_restatement_reproduces_the_reference_integrator_bitwise(path):
    """oracle_c.c runs a whole transition; with u = 0 every proposal is accepted, so
    q_out IS the integrator's end position, and E_after = V(q) + 0.5 sum(p**2) pins the
    end momentum through numpy's own sum."""
    g = load_golden(path)
    C = g['q0'].shape[0]
    k, x0 = float(g['k']), float(g['x0'])
    out = c_oracle.hmc_sample_gauss(g['q0'], g['p0'], np.zeros(C), g['timestep'], int(g['nsteps']),
                                    k=k, x0=x0)
    assert out['accepted'].all()
    assert np.array_equal(out['q_out'], g['q_out'])
    for c in range(C):
        want = 0.5 * k * np.sum((g['q_out'][c] - x0) ** 2) + 0.5 * np.sum(g['p_out'][c] ** 2)
        assert out['e_after'][c] == want


ADAPT = golden_files('ref_adapt_timestep_')


@pytest.mark.parametrize('path', ADAPT, ids=lambda p: p.split('ref_adapt_timestep_')[-1][:-4])
def test_restatements_reproduce_the_reference_adaption_bitwise(path):
    """``HMCSampler._adapt_timestep`` (hmc.py:183-191), run from the reference's own source
    (oracle/gen_ref_leapfrog.py): the step size after a sequence of accepted / rejected moves.
    The numpy restatement follows it bit for bit, and so does the C restatement's in-transition
    adaption (moves forced with u = 0 / u = inf)."""
    g = load_golden(path)
    assert 'hmc.py:183-191' in str(g['provenance'])
    flags, want = g['accepted'], g['timesteps']
    up, down, dt0 = float(g['uprate']), float(g['downrate']), float(g['timestep0'])
    C, n = flags.shape
    for c in range(C):
        s = R.RefHMCSampler(R.GaussianPDF(), np.zeros(2), dt0, 3, timestep_adaption_limit=1000,
                            adaption_uprate=up, adaption_downrate=down, variable_name='x')
        for i in range(n):
            s._last_move_accepted = bool(flags[c, i])
            s._adapt_timestep()
            assert s.timestep == want[c, i], (c, i)
    # C restatement: one chain per row, tiny steps (any proposal would be accepted at u = 0)
    q = np.zeros((C, 4))
    dts = np.full(C, dt0)
    for i in range(n):
        u = np.where(flags[:, i], 0.0, np.inf)
        out = c_oracle.hmc_sample_gauss(q, np.ones((C, 4)) * 1e-3, u, dts, 1, adapt=True, uprate=up,
                                        downrate=down)
        assert np.array_equal(out['accepted'].astype(bool), flags[:, i])
        dts = out['timestep_out']
        assert np.array_equal(dts, want[:, i]), i


SAMPLES = golden_files('ref_sample_')


@pytest.mark.parametrize('path', SAMPLES, ids=lambda p: p.split('ref_sample_')[-1][:-4])
def test_restatements_reproduce_the_reference_sample_bitwise(path):
    """``HMCSampler.sample`` (hmc.py:136-164) run from the reference's own source, statement by
    statement, except the ONE statement that needs csb (``acc = np.random.uniform() < exp(...)``,
    :151; oracle/gen_ref_leapfrog.py:_split_sample): the momentum draw, E_before, the trajectory,
    E_after, then -- with the accept flag the fixture records -- the bookkeeping, the adaption
    check AFTER the counter increment, the state replacement and the returned copy.

    The numpy and the C restatement run their WHOLE sample() (their own accept test included) on
    the recorded draws and land on the same energies, flags, states, step sizes and counters, bit
    for bit: what remains unpinned in sample() is csb's ``exp`` alone."""
    g = load_golden(path)
    assert 'every statement but :151' in str(g['provenance'])
    C, ncalls, D = g['p0'].shape
    k, x0, L = float(g['k']), float(g['x0']), int(g['nsteps'])
    limit, dt0 = int(g['adaption_limit']), float(g['timestep0'])
    for c in range(C):
        draws = {'i': 0}
        s = R.RefHMCSampler(R.GaussianPDF(k, x0), g['q0'][c].copy(), dt0, L, timestep_adaption_limit=limit,
                            variable_name='x', normal=lambda size: g['p0'][c, draws['i']].copy(),
                            uniform=lambda: g['u'][c, draws['i']])
        for i in range(ncalls):
            draws['i'] = i
            ret = s.sample()
            assert s.last_E_before == g['e_before'][c, i] and s.last_E_after == g['e_after'][c, i], (c, i)
            assert bool(s.last_move_accepted) == bool(g['accepted'][c, i])
            assert np.array_equal(s.state, g['state'][c, i]) and np.array_equal(ret, g['state'][c, i])
            assert ret is not s.state
            assert s.timestep == g['timestep'][c, i] and s.counter == g['counter'][c, i]
            assert s.n_accepted == g['n_accepted'][c, i]
    # the C restatement, all chains at once, call by call (adaption while counter < limit, hmc.py:156)
    q = g['q0'].copy()
    dts = np.full(C, dt0)
    for i in range(ncalls):
        out = c_oracle.hmc_sample_gauss(q, g['p0'][:, i], g['u'][:, i], dts, L, k=k, x0=x0,
                                        adapt=(i + 1) < limit)
        assert np.array_equal(out['accepted'].astype(bool), g['accepted'][:, i])
        assert np.array_equal(out['e_before'], g['e_before'][:, i]) and np.array_equal(out['e_after'], g['e_after'][:, i])
        assert np.array_equal(out['q_out'], g['state'][:, i])
        dts = out['timestep_out']
        assert np.array_equal(dts, g['timestep'][:, i])
        q = out['q_out']
    # the reference drew the momentum it recorded: the stream replay is consistent
    np.random.seed(int(g['seed']) + 1)
    assert np.array_equal(np.random.normal(size=D), g['p0'][0, 0]) and np.random.uniform() == g['u'][0, 0]

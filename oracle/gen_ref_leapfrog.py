"""
TEST INFRASTRUCTURE ONLY.  Writes tests/golden/ref_leapfrog_*.npz: outputs of the
REFERENCE'S OWN ``HMCSampler._leapfrog`` (``binf/samplers/hmc.py:92-125``), run in
the build container.

PROVENANCE -- read this.  ``binf/samplers/hmc.py`` cannot be imported as it stands:
its line 10 is ``from csb.numeric import exp`` and the third-party ``csb`` toolbox is
absent (not installed, not vendored).  No substitute for csb is written or
registered.  Instead the module's source is parsed, THAT ONE import statement is
dropped from the syntax tree, and the rest -- the reference's class, unchanged -- is
compiled and executed.  ``exp`` is then simply an undefined name: the only method
that uses it, ``sample()`` (``hmc.py:151``), would raise ``NameError`` and is never
called.  What IS called is ``HMCSampler.__init__`` (``:17-62``) and
``HMCSampler._leapfrog`` (``:92-125``), which touch nothing from csb: the fixtures
are outputs of the reference's integrator code, bit for bit, under numpy 2.2.6.

What this pins and what it does not:

* pinned by the reference itself: the kick-drift-kick sequence and its roundings
  (``p -= 0.5 * timestep * gradient(q)``, ``q += p * timestep``, ...), the
  ``nsteps - 1`` interior steps, the in-place update of q and p;
* the PDFs handed to it are duck-typed numpy objects written for this script (the
  reference's own PDF classes need csb): ``k * (x - x0)`` as ``binf/pdf/__init__.py:191``
  writes it, the coefficient conditional of the example (likelihood gradient only,
  quirk Q4: ``binf/pdf/likelihoods.py:148-155`` over ``binf/example/likelihood.py:24-30,
  59-61``) and the build-defined pair-distance posterior.  Their gradients are stored
  with the fixtures' inputs being enough to recompute them;
* NOT pinned: ``sample()``'s energies, accept test and the csb ``exp`` clip bounds
  (``hmc.py:136-164``) -- still "parity unpinned".

The reference never travels: only these arrays do.  Run (build container only):
    python -m oracle.gen_ref_leapfrog
"""
import ast
import os
import types

import numpy as np

REFERENCE_HMC = '/root/reference/binf/samplers/hmc.py'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')

PROVENANCE = ('outputs of the REFERENCE\'s HMCSampler._leapfrog (binf/samplers/hmc.py:92-125) '
              'executed from its own source with the single statement `from csb.numeric import '
              'exp` (hmc.py:10) dropped from the syntax tree (csb is absent; no stand-in '
              'supplied; sample(), the only user of exp, never called); numpy %s; PDFs are '
              'duck-typed numpy objects of oracle/gen_ref_leapfrog.py' % np.__version__)


def load_reference_sampler():
    """The reference's ``HMCSampler`` class, compiled from its source file minus the
    csb import."""
    with open(REFERENCE_HMC) as f:
        tree = ast.parse(f.read(), REFERENCE_HMC)
    kept, dropped = [], []
    for node in tree.body:
        if isinstance(node, ast.ImportFrom) and (node.module or '').split('.')[0] == 'csb':
            dropped.append(ast.dump(node))
        else:
            kept.append(node)
    assert len(dropped) == 1 and "'exp'" in dropped[0], dropped
    tree.body = kept
    mod = types.ModuleType('binf_reference_hmc')
    exec(compile(tree, REFERENCE_HMC, 'exec'), mod.__dict__)
    assert 'exp' not in mod.__dict__
    _split_sample(tree, mod)
    return mod.HMCSampler


def _split_sample(tree, mod):
    """``HMCSampler.sample`` (hmc.py:136-164) needs csb in ONE statement, ``acc =
    np.random.uniform() < exp(-(E_after - E_before))`` (:151).  Its other statements are
    compiled unchanged as two methods of the reference class: ``sample_until_accept(self)``
    = the statements before it (:143-150: the energy function, the copy of the state, the
    momentum draw, E_before, the leapfrog call, E_after) returning their local variables,
    and ``sample_after_accept(self, acc, q)`` = the statements after it (:153-164: the
    bookkeeping, the adaption check, the state replacement, the returned copy).  Nothing
    is written for line 151 itself: the caller draws the uniform the reference draws
    there and passes an accept flag in."""
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'HMCSampler')
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == 'sample')
    body = [n for n in fn.body if not (isinstance(n, ast.Expr) and isinstance(getattr(n, 'value', None), ast.Constant))]
    at = next(i for i, n in enumerate(body) if isinstance(n, ast.Assign) and
              getattr(n.targets[0], 'id', None) == 'acc')
    assert 'exp' in ast.dump(body[at]) and at == 6 and len(body) == 11, (at, len(body))
    before = ast.parse('def sample_until_accept(self):\n    pass').body[0]
    before.body = body[:at] + ast.parse('return E_before, E_after, q, p').body
    after = ast.parse('def sample_after_accept(self, acc, q):\n    pass').body[0]
    after.body = body[at + 1:]
    helper = ast.Module(body=[before, after], type_ignores=[])
    ast.fix_missing_locations(helper)
    exec(compile(helper, REFERENCE_HMC + ':sample[split at :151]', 'exec'), mod.__dict__)
    mod.HMCSampler.sample_until_accept = mod.__dict__.pop('sample_until_accept')
    mod.HMCSampler.sample_after_accept = mod.__dict__.pop('sample_after_accept')


# -- duck-typed PDFs (numpy, one chain) ------------------------------------------------------
class Gaussian(object):
    """gradient of the energy as binf/pdf/__init__.py:191 writes it."""

    def __init__(self, k, x0):
        self.k, self.x0 = k, x0

    def gradient(self, x):
        return self.k * (x - self.x0)

    def log_prob(self, x):                      # binf/pdf/__init__.py:185
        return -0.5 * self.k * np.sum((x - self.x0) ** 2)


class PolyCoefficients(object):
    """The force on ``coefficients`` in the example's conditional posterior: the
    likelihood alone (the prior registers its variable as non-differentiable, Q4):
    jacobi_matrix.dot(error-model gradient), binf/pdf/likelihoods.py:148-155."""

    def __init__(self, xses, ys, precision):
        self.xses, self.ys, self.precision = xses, ys, precision

    def gradient(self, coefficients):
        mock = np.polynomial.polynomial.polyval(self.xses, coefficients)     # likelihood.py:26
        emgrad = (mock - self.ys) * self.precision                           # likelihood.py:59-61
        dfm = np.vstack([self.xses ** i for i in range(len(coefficients))])  # likelihood.py:28-30
        return dfm.dot(emgrad)                                               # likelihoods.py:155


class Distance(object):
    """Build-defined pair-distance restraint posterior (oracle/ref_distance.py): Gaussian
    error model on all pair distances + isotropic Gaussian prior k."""

    def __init__(self, ys, precision, n_beads, prior_k):
        from oracle import ref_distance
        self._pdf = ref_distance.DistancePosterior(ys, precision, n_beads, prior_k=prior_k)

    def gradient(self, coordinates):
        return self._pdf.gradient(coordinates=coordinates)


def run_cases(Sampler, make_pdf, name, q0, p0, dt, L):
    """q0, p0: [C x D]; dt: float or [C]; one reference sampler per chain, as the reference
    runs (one chain per object)."""
    C = q0.shape[0]
    q_out, p_out = np.empty_like(q0), np.empty_like(p0)
    for c in range(C):
        s = Sampler(make_pdf(), q0[c].copy(), float(np.asarray(dt).reshape(-1)[c % np.size(dt)]), L,
                    variable_name=name)
        q, p = q0[c].copy(), p0[c].copy()
        rq, rp = s._leapfrog(q, p, s.timestep, s.nsteps)
        assert rq is q and rp is p                        # in place (hmc.py:116-125)
        q_out[c], p_out[c] = q, p
    return q_out, p_out


def main():
    Sampler = load_reference_sampler()
    os.makedirs(OUT, exist_ok=True)
    written = []

    def save(tag, **arrays):
        path = os.path.join(OUT, 'ref_leapfrog_%s.npz' % tag)
        np.savez_compressed(path, provenance=np.array(PROVENANCE), **arrays)
        written.append(os.path.basename(path))

    # Gaussian (the TestHO form), SURVEY 8(c)'s sets: D in {4, 33, 768, 1024}, L in {1, 2, 20, 50}
    for tag, D, L, k, x0, dt, C, seed in [
            ('gauss_d4_l1', 4, 1, 1.0, 0.0, 0.30, 8, 100),
            ('gauss_d4_l50_k2p5', 4, 50, 2.5, 0.3, 0.20, 8, 110),
            ('gauss_d7_l2', 7, 2, 1.0, 0.0, 0.90, 8, 120),
            ('gauss_d33_l20_k2p5', 33, 20, 2.5, 0.3, 0.35, 8, 130),
            ('gauss_d200_l20', 200, 20, 1.0, -0.2, 0.30, 4, 150),
            ('gauss_d768_l20', 768, 20, 1.0, 0.0, 0.22, 4, 160),
            ('gauss_d1024_l20', 1024, 20, 1.0, 0.0, 0.05, 6, 170),
            ('gauss_d1024_l50_k2p5', 1024, 50, 2.5, 0.3, 0.11, 3, 180),
            ('gauss_d9000_l3', 9000, 3, 1.0, 0.0, 0.40, 2, 190)]:
        rs = np.random.RandomState(seed)
        q0 = rs.standard_normal((C, D))
        p0 = rs.standard_normal((C, D))
        q, p = run_cases(Sampler, lambda: Gaussian(k, x0), 'x', q0, p0, dt, L)
        save(tag, kind=np.array('gauss'), q0=q0, p0=p0, q_out=q, p_out=p, timestep=np.float64(dt),
             nsteps=np.int64(L), k=np.float64(k), x0=np.float64(x0))
    # per-chain step sizes (what the adaption leaves behind, hmc.py:183-191)
    rs = np.random.RandomState(200)
    q0, p0 = rs.standard_normal((6, 33)), rs.standard_normal((6, 33))
    dts = 0.3 * 1.05 ** np.arange(6) * 0.95 ** np.arange(6)[::-1]
    q, p = run_cases(Sampler, lambda: Gaussian(1.0, 0.0), 'x', q0, p0, dts, 5)
    save('gauss_d33_l5_dtchain', kind=np.array('gauss'), q0=q0, p0=p0, q_out=q, p_out=p, timestep=dts,
         nsteps=np.int64(5), k=np.float64(1.0), x0=np.float64(0.0))

    # the example's coefficient conditional: example_script.py:17-26 (K = 4, N = 20, L = 50) and
    # the C3 shape (K = 33, N = 16384, xs in [-1, 1])
    np.random.seed(0)
    xs = np.linspace(-2, 2, 20)
    ys = np.random.normal(loc=np.polynomial.polynomial.polyval(xs, np.array([2.0, -4.0, 1.0, 1.5])),
                          scale=1.0 / np.sqrt(2.5))
    rs = np.random.RandomState(300)
    q0 = np.ones((5, 4)) + 0.1 * rs.standard_normal((5, 4))
    p0 = rs.standard_normal((5, 4))
    q, p = run_cases(Sampler, lambda: PolyCoefficients(xs, ys, 1.0), 'coefficients', q0, p0, 0.02, 50)
    save('poly_k4_n20_l50', kind=np.array('poly'), q0=q0, p0=p0, q_out=q, p_out=p, timestep=np.float64(0.02),
         nsteps=np.int64(50), xs=xs, ys=ys, precision=np.float64(1.0))
    K, N = 33, 16384
    xs = np.linspace(-1, 1, N)
    ys = np.polynomial.polynomial.polyval(xs, np.random.RandomState(7).standard_normal(K)) + \
        np.random.RandomState(9).standard_normal(N) / np.sqrt(2.5)
    rs = np.random.RandomState(310)
    q0, p0 = rs.standard_normal((3, K)), rs.standard_normal((3, K))
    q, p = run_cases(Sampler, lambda: PolyCoefficients(xs, ys, 2.5), 'coefficients', q0, p0, 2e-4, 20)
    save('poly_k33_n16384_l20', kind=np.array('poly'), q0=q0, p0=p0, q_out=q, p_out=p,
         timestep=np.float64(2e-4), nsteps=np.int64(20), xs=xs, ys=ys, precision=np.float64(2.5))

    # pair-distance restraint posterior (build-defined model; the integrator is the reference's)
    for tag, n, C, L, dt, prior_k, seed in [('dist_n12_l10', 12, 4, 10, 0.01, 0.05, 400),
                                            ('dist_n64_l20', 64, 3, 20, 0.004, 0.05, 410),
                                            ('dist_n100_l5_noprior', 100, 2, 5, 0.003, 0.0, 420)]:
        rs = np.random.RandomState(seed)
        truth = rs.standard_normal((n, 3)) * 2.0
        I, J = np.triu_indices(n, 1)
        d = np.sqrt(np.sum((truth[I] - truth[J]) ** 2, axis=1))
        ys = np.abs(d + 0.05 * rs.standard_normal(len(d)))
        q0 = truth.reshape(-1)[None, :] + 0.1 * rs.standard_normal((C, 3 * n))
        p0 = rs.standard_normal((C, 3 * n))
        q, p = run_cases(Sampler, lambda: Distance(ys, 4.0, n, prior_k), 'coordinates', q0, p0, dt, L)
        save(tag, kind=np.array('dist'), q0=q0, p0=p0, q_out=q, p_out=p, timestep=np.float64(dt),
             nsteps=np.int64(L), ys=ys, precision=np.float64(4.0), n_beads=np.int64(n),
             prior_k=np.float64(prior_k))
    # HMCSampler._adapt_timestep (hmc.py:183-191) is free of csb as well: the step size after a
    # sequence of accepted / rejected moves, in the reference's own order of multiplications
    # (uprate on ACCEPT, quirk Q3).  The flags are set the way sample() sets them (hmc.py:153).
    for tag, dt0, up, down, seed in [('default_rates', 0.3, 1.05, 0.95, 500), ('other_rates', 0.0123, 1.3, 0.6, 510)]:
        flags = np.random.RandomState(seed).uniform(size=(16, 12)) < 0.7
        steps = np.empty(flags.shape)
        for c in range(flags.shape[0]):
            s = Sampler(Gaussian(1.0, 0.0), np.zeros(2), dt0, 3, timestep_adaption_limit=1000,
                        adaption_uprate=up, adaption_downrate=down, variable_name='x')
            for i, f in enumerate(flags[c]):
                s._last_move_accepted = bool(f)
                s._adapt_timestep()
                steps[c, i] = s.timestep
        path = os.path.join(OUT, 'ref_adapt_timestep_%s.npz' % tag)
        np.savez_compressed(path, provenance=np.array(PROVENANCE.replace(
            "HMCSampler._leapfrog (binf/samplers/hmc.py:92-125)",
            "HMCSampler._adapt_timestep (binf/samplers/hmc.py:183-191)")), kind=np.array('adapt'),
            accepted=flags, timestep0=np.float64(dt0), uprate=np.float64(up), downrate=np.float64(down),
            timesteps=steps)
        written.append(os.path.basename(path))
    # ---- sample() itself, every statement but the one that needs csb (see _split_sample) ------
    for tag, D, L, k, x0, dt, C, ncalls, limit, seed in [
            ('gauss_d4_l1', 4, 1, 1.0, 0.0, 0.30, 8, 3, 0, 600),
            ('gauss_d33_l20_k2p5', 33, 20, 2.5, 0.3, 0.35, 8, 3, 0, 610),
            ('gauss_d33_l2_adapt', 33, 2, 1.0, 0.0, 0.70, 8, 5, 4, 620),
            ('gauss_d200_l20', 200, 20, 1.0, -0.2, 0.30, 4, 2, 0, 630),
            ('gauss_d1024_l20', 1024, 20, 1.0, 0.0, 0.22, 6, 3, 0, 640),
            ('gauss_d1024_l5_adapt', 1024, 5, 2.5, 0.3, 0.30, 4, 4, 10, 650),
            ('gauss_d9000_l3', 9000, 3, 1.0, 0.0, 0.02, 2, 2, 0, 660)]:
        q0 = np.random.RandomState(seed).standard_normal((C, D))
        rec = {n: [] for n in ('p0', 'u', 'e_before', 'e_after', 'accepted', 'state', 'timestep',
                               'counter', 'n_accepted', 'proposal')}
        for c in range(C):
            np.random.seed(seed + 1 + c)                    # chain c consumes its own legacy stream
            s = Sampler(Gaussian(k, x0), q0[c].copy(), dt, L, timestep_adaption_limit=limit, variable_name='x')
            row = {n: [] for n in rec}
            for _ in range(ncalls):
                stream = np.random.get_state()
                e_b, e_a, q, p = s.sample_until_accept()                       # hmc.py:143-150
                shadow = np.random.RandomState()
                shadow.set_state(stream)
                row['p0'].append(shadow.normal(size=D))                       # the draw of :146, replayed
                u = np.random.uniform()                                        # the draw of :151
                # the accept flag: the ONE thing not computed by reference code (csb's exp is absent);
                # numpy's exp of the reference's own energies (|dE| is far inside any clip bound here)
                acc = bool(u < np.exp(-(e_a - e_b)))
                ret = s.sample_after_accept(acc, q)                            # hmc.py:153-164
                assert ret is not s.state and np.array_equal(ret, s.state)    # a copy (Q9)
                for n, v in (('u', u), ('e_before', e_b), ('e_after', e_a), ('accepted', acc),
                             ('state', s.state.copy()), ('timestep', s.timestep), ('counter', s.counter),
                             ('n_accepted', s.n_accepted), ('proposal', q.copy())):
                    row[n].append(v)
            for n in rec:
                rec[n].append(np.array(row[n]))
        path = os.path.join(OUT, 'ref_sample_%s.npz' % tag)
        np.savez_compressed(
            path, provenance=np.array(PROVENANCE.replace(
                "HMCSampler._leapfrog (binf/samplers/hmc.py:92-125)",
                "HMCSampler.sample (binf/samplers/hmc.py:136-164), every statement but :151 -- split there into "
                "the statements before (:143-150) and after (:153-164); the accept flag passed in is u < "
                "numpy.exp(-(E_after - E_before)) of the reference's own energies, computed by the generating "
                "script --") + '; chain c: np.random.seed(seed + 1 + c)'),
            kind=np.array('gauss'), q0=q0, k=np.float64(k), x0=np.float64(x0), timestep0=np.float64(dt),
            nsteps=np.int64(L), adaption_limit=np.int64(limit), seed=np.int64(seed),
            **{n: np.array(v) for n, v in rec.items()})                        # [C, ncalls, ...]
        written.append(os.path.basename(path))

    # the csb-free attribute plumbing of the reference's class (hmc.py:56-90,127-134,166-181):
    # what GibbsSampler and user code read from a sampler
    import json
    s = Sampler(Gaussian(1.0, 0.0), np.arange(3.0), 0.25, 7)             # variable_name left at None
    fresh = {'acceptance_rate': s.acceptance_rate, 'variable_name': s.variable_name,
             'last_move_accepted': s.last_move_accepted, 'n_accepted': s.n_accepted, 'counter': s.counter,
             'timestep': s.timestep, 'nsteps': s.nsteps,
             'timestep_adaption_limit': s.timestep_adaption_limit,
             'adaption_uprate': s.adaption_uprate, 'adaption_downrate': s.adaption_downrate}
    stats = s.last_draw_stats
    fresh['last_draw_stats'] = {k: {'fields': list(v._fields), 'values': list(v)} for k, v in stats.items()}
    s.n_accepted, s.counter = 3, 4
    s._last_move_accepted = True
    named = Sampler(Gaussian(1.0, 0.0), np.arange(3.0), 0.25, 7, variable_name='coefficients')
    copy = s._copy_state(s.state)
    plumbing = {'provenance': PROVENANCE.replace("HMCSampler._leapfrog (binf/samplers/hmc.py:92-125)",
                                                 "HMCSampler attribute plumbing (binf/samplers/hmc.py:56-90,127-134,166-181)"),
                'fresh': fresh,
                'after_3_of_4': {'acceptance_rate': s.acceptance_rate, 'last_move_accepted': s.last_move_accepted,
                                 'last_draw_stats_values': list(s.last_draw_stats['HMC'])},
                'named': {'variable_name': named.variable_name, 'last_draw_stats_keys': list(named.last_draw_stats)},
                'copy_state': {'equal': bool(np.array_equal(copy, s.state)), 'same_object': copy is s.state}}
    # BinfState (binf/samplers/__init__.py:9-57) is a plain object class too: the module's two
    # csb import statements (:5-6, names BinfState never touches) are dropped the same way
    with open('/root/reference/binf/samplers/__init__.py') as f:
        tree = ast.parse(f.read())
    n_before = len(tree.body)
    tree.body = [n for n in tree.body if not (isinstance(n, ast.ImportFrom) and (n.module or '').split('.')[0] == 'csb')]
    assert n_before - len(tree.body) == 2
    smod = types.ModuleType('binf_reference_samplers')
    exec(compile(tree, 'binf/samplers/__init__.py', 'exec'), smod.__dict__)
    st = smod.BinfState({'b': 2.0, 'a': 1.0})
    view = st.variables
    view['a'] = 99.0                                    # a COPY: the state must not change
    st.update_variables(c=3.0, a=1.5)
    st.update_momenta(a=-1.0)
    plumbing['binf_state'] = {'variables_after_update': dict(st.variables), 'copy_is_detached': st.variables['a'] == 1.5,
                              'view_after_write': view, 'momenta': dict(st.momenta),
                              'fresh_is_empty': smod.BinfState().variables == {} and smod.BinfState().momenta == {},
                              'provenance': 'BinfState of binf/samplers/__init__.py:9-57 executed from its source with '
                                            'the two csb import statements (:5-6) dropped; nothing substituted'}
    path = os.path.join(OUT, 'ref_hmc_attributes.json')
    with open(path, 'w') as f:
        json.dump(plumbing, f, indent=1, sort_keys=True)
    written.append(os.path.basename(path))
    print('wrote %d files to %s:\n  %s' % (len(written), OUT, '\n  '.join(written)))


if __name__ == '__main__':
    main()

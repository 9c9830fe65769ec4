"""
TEST INFRASTRUCTURE ONLY.  Writes the golden vectors under tests/golden/.

PROVENANCE -- read this: the vectors are outputs of the numpy RESTATEMENT in
oracle/ref_numpy.py, NOT of the reference.  The reference package cannot be
imported in the build container (it needs the absent third-party ``csb``
toolbox) and no substitute module is fabricated to force that import.  For the
HMC numerics these fixtures therefore carry the status **parity unpinned**;
they freeze the restatement so that the C oracle, the HIP kernels and later
rounds are all held to the same bits.  Each .npz stores that statement in its
``provenance`` field.

RNG-stream consumption follows binf/samplers/hmc.py:146,151: per sample() one
``np.random.normal(size=D)`` then one ``np.random.uniform()`` from the global
legacy stream; chain c of a set with seed s uses ``np.random.seed(s + c)``.

Run:  python -m oracle.gen_golden
"""
import os

import numpy as np

from oracle import ref_numpy as R

PROVENANCE = ("outputs of oracle/ref_numpy.py (restatement of "
              "binf/samplers/hmc.py:92-164 + binf/pdf/__init__.py:181-191); "
              "reference not importable (csb absent) -> parity unpinned")

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')

# name, D, L, k, x0, dt, C, ncalls, adaption_limit, seed
GAUSS_SETS = [
    ('gauss_d4_l1',       4,    1,  1.0, 0.0,  0.30,  8, 3, 0, 100),
    ('gauss_d4_l50_k2p5', 4,    50, 2.5, 0.3,  0.20,  8, 3, 0, 110),
    ('gauss_d7_l2',       7,    2,  1.0, 0.0,  0.90,  8, 2, 0, 120),
    ('gauss_d33_l20',     33,   20, 2.5, 0.3,  0.35,  8, 3, 0, 130),
    ('gauss_d33_l2_adapt', 33,  2,  1.0, 0.0,  0.70,  8, 4, 3, 140),
    ('gauss_d200_l20',    200,  20, 1.0, -0.2, 0.30,  4, 2, 0, 150),
    ('gauss_d768_l20',    768,  20, 1.0, 0.0,  0.22,  4, 2, 0, 160),
    ('gauss_d768_l50_k2p5', 768, 50, 2.5, 0.3, 0.12,  4, 2, 0, 170),
    ('gauss_d1023_l2',    1023, 2,  1.0, 0.0,  0.25,  4, 2, 0, 180),
    ('gauss_d1024_l20',   1024, 20, 1.0, 0.0,  0.05,  4, 2, 0, 190),
    ('gauss_d1024_l20_bigdt', 1024, 20, 1.0, 0.0, 0.20, 6, 2, 0, 200),
    ('gauss_d1024_l1_adapt', 1024, 1, 2.5, 0.3, 0.18, 4, 4, 4, 210),
    # longer than numpy's 8192-element reduction buffer (one full chunk + a ragged one)
    ('gauss_d8200_l2_adapt', 8200, 2, 1.0, 0.0, 0.16, 2, 3, 3, 220),
]


def run_gauss_set(D, L, k, x0, dt, C, ncalls, limit, seed):
    q0 = np.random.RandomState(seed + 7919).standard_normal((C, D))
    p0 = np.empty((ncalls, C, D))
    u = np.empty((ncalls, C))
    q_out = np.empty((ncalls, C, D))
    acc = np.empty((ncalls, C), dtype=np.uint8)
    eb = np.empty((ncalls, C))
    ea = np.empty((ncalls, C))
    dt_out = np.empty((ncalls, C))
    for c in range(C):
        np.random.seed(seed + c)
        s = R.RefHMCSampler(R.GaussianPDF(k, x0), q0[c].copy(), dt, L,
                            timestep_adaption_limit=limit, variable_name='x')
        for i in range(ncalls):
            st = np.random.get_state()
            p0[i, c] = np.random.normal(size=D)
            u[i, c] = np.random.uniform()
            np.random.set_state(st)          # sample() consumes the same draws
            q_out[i, c] = s.sample()
            acc[i, c] = 1 if s.last_move_accepted else 0
            eb[i, c] = s.last_E_before
            ea[i, c] = s.last_E_after
            dt_out[i, c] = s.timestep
    return dict(q0=q0, p0=p0, u=u, q_out=q_out, accepted=acc, e_before=eb,
                e_after=ea, timestep_out=dt_out)


# name, K, N, C, sweeps, L, dt, seed  -- Gibbs-within-HMC on the polynomial model
# (BASELINE C1 / C4 at CPU size).  'poly_c1_example' is example_script.py:17-26
# itself (np.random.seed(0) data, start coefficients = ones, precision = 1).
POLY_SETS = [
    ('poly_c1_example', 4, 20, 4, 5, 50, 0.02, 300),
    ('poly_k7_n37',     7, 37, 3, 3, 10, 0.004, 310),
    ('poly_k16_n128',  16, 128, 2, 2, 4, 0.0005, 320),
    ('poly_k33_n16384', 33, 16384, 2, 2, 5, 1e-6, 330),      # BASELINE C3's shape (per-step tier)
]

# name, n_beads, C, ncalls, L, dt, precision, prior_k, seed  -- pair-distance model (C5, build-defined)
DIST_SETS = [
    ('dist_n12', 12, 4, 2, 10, 0.01, 4.0, 0.05, 400),
    ('dist_n40', 40, 2, 2, 5, 0.004, 2.0, 0.0, 410),
    # round 2: the every-pair-once force scheme (4 / 16 waves) and the size above it
    ('dist_n100', 100, 2, 2, 4, 0.002, 3.0, 0.05, 420),
    ('dist_n256', 256, 2, 2, 3, 0.001, 4.0, 0.05, 430),      # BASELINE C5's bead count
    ('dist_n300', 300, 1, 2, 2, 0.001, 4.0, 0.0, 440),
]


def run_poly_set(K, N, C, sweeps, L, dt, seed):
    """Per chain c: np.random.seed(seed + c); every sweep consumes the global
    legacy stream in the reference's order -- normal(size=K), uniform()
    (HMCSampler.sample, hmc.py:146,151), then gamma(shape) (GammaSampler.sample,
    binf/example/samplers.py:47)."""
    from oracle import ref_example as RE
    if K == 4 and N == 20:
        xs, ys = RE.example_data()
        c0 = np.ones((C, K))
        t0 = np.ones(C)
    else:
        rs = np.random.RandomState(seed + 7919)
        xs = np.linspace(-1.5, 1.5, N)
        ys = R.polyval(xs, rs.standard_normal(K)) + 0.5 * rs.standard_normal(N)
        c0 = 0.2 * rs.standard_normal((C, K))
        t0 = rs.uniform(0.5, 2.0, size=C)
    shape = R.gamma_shape(N, 1.0)
    p0 = np.empty((sweeps, C, K))
    u = np.empty((sweeps, C))
    g = np.empty((sweeps, C))
    out = {k: [] for k in ('coefficients', 'precision', 'accepted', 'e_before', 'e_after')}
    for c in range(C):
        np.random.seed(seed + c)
        for s in range(sweeps):
            p0[s, c] = np.random.normal(size=K)
            u[s, c] = np.random.uniform()
            g[s, c] = np.random.gamma(shape)
        r = RE.gibbs_hmc_chain(xs, ys, c0[c], t0[c], dt, L, p0[:, c], u[:, c], g[:, c])
        for k in out:
            out[k].append(r[k])
    res = {k: np.stack(v, axis=1) for k, v in out.items()}       # [sweeps, C, ...]
    res['accepted'] = res['accepted'].astype(np.uint8)
    return dict(xs=xs, ys=ys, coefficients0=c0, precision0=t0, p0=p0, u=u, gamma=g,
                gamma_shape=shape, **res)


def run_dist_set(n, C, ncalls, L, dt, precision, prior_k, seed):
    from oracle import ref_distance as RD
    rs = np.random.RandomState(seed + 7919)
    truth = rs.standard_normal((n, 3)) * 2.0
    ys = np.abs(RD.forward(truth.reshape(-1), n) + 0.05 * rs.standard_normal(n * (n - 1) // 2))
    x0 = truth.reshape(-1)[None, :] + 0.3 * rs.standard_normal((C, 3 * n))
    D = 3 * n
    p0 = np.empty((ncalls, C, D))
    u = np.empty((ncalls, C))
    q_out = np.empty((ncalls, C, D))
    acc = np.empty((ncalls, C), dtype=np.uint8)
    eb = np.empty((ncalls, C))
    ea = np.empty((ncalls, C))
    logp = np.array([RD.log_prob(x0[c], ys, precision, n) for c in range(C)])
    grad = np.stack([RD.gradient(x0[c], ys, precision, n) for c in range(C)])
    for c in range(C):
        np.random.seed(seed + c)
        s = R.RefHMCSampler(RD.DistancePosterior(ys, precision, n, prior_k=prior_k), x0[c].copy(),
                            dt, L, variable_name='coordinates')
        for i in range(ncalls):
            st = np.random.get_state()
            p0[i, c] = np.random.normal(size=D)
            u[i, c] = np.random.uniform()
            np.random.set_state(st)
            q_out[i, c] = s.sample()
            acc[i, c] = 1 if s.last_move_accepted else 0
            eb[i, c] = s.last_E_before
            ea[i, c] = s.last_E_after
    return dict(ys=ys, q0=x0, p0=p0, u=u, q_out=q_out, accepted=acc, e_before=eb, e_after=ea,
                likelihood_log_prob=logp, likelihood_gradient=grad)


def main(only=None):
    """only: write just the sets whose name is in that collection (adding a
    fixture must not rewrite the existing files)."""
    os.makedirs(OUT, exist_ok=True)
    for name, K, N, C, sweeps, L, dt, seed in POLY_SETS:
        if only is not None and name not in only:
            continue
        r = run_poly_set(K, N, C, sweeps, L, dt, seed)
        np.savez(os.path.join(OUT, name + '.npz'), K=K, N=N, L=L, timestep=dt, seed=seed,
                 provenance=PROVENANCE.replace('binf/samplers/hmc.py:92-164 + binf/pdf/__init__.py:181-191',
                                               'hmc.py, gibbs.py, posteriors.py, likelihoods.py, binf/example/*'),
                 **r)
        print('%-26s acc=%s' % (name, r['accepted'].mean(axis=1)))
    for name, n, C, ncalls, L, dt, precision, prior_k, seed in DIST_SETS:
        if only is not None and name not in only:
            continue
        r = run_dist_set(n, C, ncalls, L, dt, precision, prior_k, seed)
        np.savez(os.path.join(OUT, name + '.npz'), n_beads=n, L=L, timestep=dt, precision=precision,
                 prior_k=prior_k, seed=seed,
                 provenance='outputs of oracle/ref_distance.py + ref_numpy.py; the model is '
                            'build-defined (no reference code exists) -> parity unpinned',
                 **r)
        print('%-26s acc=%s' % (name, r['accepted'].mean(axis=1)))
    for name, D, L, k, x0, dt, C, ncalls, limit, seed in GAUSS_SETS:
        if only is not None and name not in only:
            continue
        r = run_gauss_set(D, L, k, x0, dt, C, ncalls, limit, seed)
        np.savez(os.path.join(OUT, name + '.npz'), D=D, L=L, k=k, x0=x0,
                 timestep=dt, adaption_limit=limit, seed=seed,
                 uprate=1.05, downrate=0.95, provenance=PROVENANCE, **r)
        print('%-26s acc=%s' % (name, r['accepted'].mean(axis=1)))


if __name__ == '__main__':
    import sys
    main(only=set(sys.argv[1:]) or None)

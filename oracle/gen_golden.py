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


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, D, L, k, x0, dt, C, ncalls, limit, seed in GAUSS_SETS:
        r = run_gauss_set(D, L, k, x0, dt, C, ncalls, limit, seed)
        np.savez(os.path.join(OUT, name + '.npz'), D=D, L=L, k=k, x0=x0,
                 timestep=dt, adaption_limit=limit, seed=seed,
                 uprate=1.05, downrate=0.95, provenance=PROVENANCE, **r)
        print('%-26s acc=%s' % (name, r['accepted'].mean(axis=1)))


if __name__ == '__main__':
    main()

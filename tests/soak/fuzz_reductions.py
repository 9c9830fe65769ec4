"""Randomised differential test of the np.sum-order reductions behind the
generic tier and the polynomial log-prob, bit for bit against numpy.
Development aid / soak test:  python tests/soak/fuzz_reductions.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from binf_amd import _native

dev = torch.device('cuda:0')
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
polyval = np.polynomial.polynomial.polyval
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
bad = 0
t0 = time.time()
for case in range(n_cases):
    kind = rs.randint(4)
    D = int([rs.randint(0, 300), rs.randint(300, 8193), rs.randint(7600, 8300),
             rs.randint(8193, 60000)][kind])
    C = int(rs.choice([1, 2, 5, 33]))
    x = rs.standard_normal((C, D)) * 10 ** rs.uniform(-3, 3)
    y = rs.standard_normal(D)
    w = rs.uniform(0.5, 2.0, D)
    ok = np.array_equal(_native.row_sum(t(x)).cpu().numpy(), np.array([np.sum(r) for r in x]))
    ok &= np.array_equal(_native.row_sum(t(x), _native.ROW_SUMSQ, scale=0.5).cpu().numpy(),
                         np.array([0.5 * np.sum(r ** 2) for r in x]))
    ok &= np.array_equal(_native.row_sumsq_diff(t(x), t(y), scale=-0.5, weights=t(w)).cpu().numpy(),
                         np.array([-0.5 * np.sum((r - y) ** 2 / w) for r in x]))
    if D > 0:
        lp = rs.standard_normal(C) * 10 ** rs.uniform(-2, 4)
        ok &= np.array_equal(_native.hmc_energy(t(x), t(lp)).cpu().numpy(),
                             np.array([-lp[c] + 0.5 * np.sum(x[c] ** 2) for c in range(C)]))
    if not ok:
        bad += 1
        print('MISMATCH row reduction', dict(D=D, C=C), flush=True)
    # polynomial chi^2 (precision 1: the log term vanishes, the rest is exact)
    K = int(rs.randint(1, 65))
    N = int([rs.randint(0, 200), rs.randint(200, 8193), rs.randint(7600, 8300),
             rs.randint(8193, 40000)][rs.randint(4)])
    Cp = int(rs.choice([1, 3, 17]))
    xs = np.linspace(-1, 1, N) if N else np.zeros(0)
    ys = rs.standard_normal(N)
    th = rs.standard_normal((Cp, K))
    got = _native.poly_gauss_logp(t(th), t(xs), t(ys), 1.0).cpu().numpy()
    want = np.array([-0.5 * np.sum((polyval(xs, c) - ys) ** 2) * 1.0 + N * 0.5 * np.log(1.0)
                     for c in th])
    if N and not np.array_equal(got, want):
        bad += 1
        print('MISMATCH poly logp', dict(K=K, N=N, C=Cp), flush=True)
    if case % 50 == 49:
        print('%d cases, %d mismatches, %.0f s' % (case + 1, bad, time.time() - t0), flush=True)
print('done: %d cases, %d mismatches' % (n_cases, bad))
sys.exit(1 if bad else 0)

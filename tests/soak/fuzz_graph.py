"""Randomised differential test of ``HMCSampler(graph=True)``: user-style torch PDFs and library
PDFs on the per-step tier, random shapes / step counts / adaption limits / modes; every call of
the graphed sampler against the eager sampler, bit for bit.  Also shows that many captures in
one process neither leak nor fail.
Development aid / soak test:  python tests/soak/fuzz_graph.py [n_cases] [seed]"""
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler

dev = torch.device('cuda:0')
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


class DoubleWell(object):
    def __init__(self, a):
        self.a = a

    def log_prob(self, x):
        w = x * x - 1.0
        return (-self.a) * (w * w).sum(dim=1)

    def gradient(self, x):
        return (4.0 * self.a) * x * (x * x - 1.0)


class Quartic(object):
    """log p = -sum(b x^4 + 0.5 x^2), with a device tensor of per-dimension weights"""

    def __init__(self, b):
        self.b = b

    def log_prob(self, x):
        return -((self.b * x ** 4).sum(dim=1) + 0.5 * (x * x).sum(dim=1))

    def gradient(self, x):
        return 4.0 * self.b * x ** 3 + x


t0 = time.time()
bad = 0
for case in range(n_cases):
    C = int(rs.choice([1, 2, 7, 33, 64, 200, 1000]))
    D = int(rs.choice([1, 3, 16, 33, 100, 257, 1024]))
    if C * D > 300000:
        C = max(1, 300000 // D)
    L = int(rs.randint(1, 8))
    n = int(rs.randint(3, 9))
    limit = int(rs.choice([0, 0, 3, 100]))
    mode = 'fma' if rs.rand() < 0.3 else 'exact'
    kind = rs.randint(3)
    def make():
        if kind == 0:
            return DoubleWell(float(1.5))
        if kind == 1:
            return Quartic(torch.linspace(0.1, 0.5, D, dtype=torch.float64, device=dev))
        pdf = IsotropicGaussian(2.5, 0.3)
        pdf.native_hmc_spec = lambda name: None
        return pdf
    q0 = torch.from_numpy(rs.standard_normal((C, D)) * 0.5).to(dev)
    p0 = torch.from_numpy(rs.standard_normal((n, C, D))).to(dev)
    u = torch.from_numpy(rs.uniform(size=(n, C))).to(dev)
    dt = float(rs.uniform(0.02, 0.3))
    kw = dict(variable_name='x', timestep_adaption_limit=limit, mode=mode, record_energies=True)
    e = HMCSampler(make(), q0.clone(), dt, L, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter('error')                 # graph mode must not give up here
        g = HMCSampler(make(), q0.clone(), dt, L, graph='always', **kw)
        ok = True
        for i in range(n):
            a = e.sample(p0=p0[i], u=u[i])
            b = g.sample(p0=p0[i], u=u[i])
            ok &= torch.equal(a, b) and torch.equal(e.last_move_accepted, g.last_move_accepted)
            ok &= torch.equal(e.last_e_before, g.last_e_before) and torch.equal(e.last_e_after, g.last_e_after)
    ok &= torch.equal(e.n_accepted, g.n_accepted) and len(g._graphs) >= 1
    if limit:
        ok &= torch.equal(torch.as_tensor(e.timestep), torch.as_tensor(g.timestep))
    if not ok:
        bad += 1
        print('MISMATCH', dict(C=C, D=D, L=L, n=n, limit=limit, mode=mode, kind=int(kind)), flush=True)
    del g, e
    if case % 50 == 49:
        print('%d cases, %d mismatches, %.0f s, %.0f MiB reserved' % (
            case + 1, bad, time.time() - t0, torch.cuda.memory_reserved() / 2 ** 20), flush=True)
print('done: %d cases, %d mismatches' % (n_cases, bad))
sys.exit(1 if bad else 0)

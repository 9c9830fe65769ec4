"""Fused pair-distance leapfrog (L = 20) for small bead counts and many chains: the
every-pair-once scheme against the one-sided loops (BINF_PD_SYM=0), development aid."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
dev = torch.device('cuda:0')
rs = np.random.RandomState(0)
for n in (8, 24, 48, 64, 100, 128, 200, 256):
    truth = rs.standard_normal((n, 3)) * 2.0
    d = np.sqrt(((truth[:, None, :] - truth[None, :, :]) ** 2).sum(-1))
    ymat = torch.from_numpy(np.abs(d + 0.05 * rs.standard_normal((n, n)))).to(dev)
    row = []
    for C in (256, 2048, 16384):
        q = torch.from_numpy(truth.reshape(-1)[None, :] + 0.1 * rs.standard_normal((C, 3 * n))).to(dev)
        p = torch.zeros_like(q)
        f = lambda: _native.pairdist_leapfrog(q, p, ymat, 4.0, (0.01, 0.0), True, 1e-5, None, 20)
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        row.append('C=%d %.0f us' % (C, e0.elapsed_time(e1) * 1e3 / 20))
    print('sym=%s n=%3d  %s' % (os.environ.get('BINF_PD_SYM', '1'), n, '  '.join(row)), flush=True)

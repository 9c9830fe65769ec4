#!/usr/bin/env python3
"""A/B of two builds of the MFMA gradient kernel (BINF_LIB_OVERRIDE selects the library):
time per launch at C3 / C4 shapes and a checksum of the result (bitwise comparison between runs)."""
import hashlib, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
from binf_amd.example.likelihood import POLYVAL, ForwardModel
dev = torch.device('cuda:0')
out = {}
for C, K, N in ((8192, 33, 16384), (4096, 33, 16384), (130, 33, 1000), (2100, 17, 50), (64, 4, 20), (300, 64, 700),
                # whole tiles (N % 16 == 0): the trimmed kernel unless BINF_POLY_GRAD_GENERAL=1
                (130, 33, 1024), (2100, 17, 64), (64, 4, 32), (300, 64, 704), (300, 48, 1600), (1500, 50, 320),
                (8192, 16, 4096), (2050, 34, 16), (5, 36, 48), (8192, 32, 16384), (1024, 33, 16384)):
    xs = np.linspace(-1, 1, N)
    ys = np.random.RandomState(9).standard_normal(N)
    q0 = torch.from_numpy(np.random.RandomState(8).standard_normal((C, K))).to(dev)
    A = ForwardModel(xs, POLYVAL).design_matrix(K, dev)
    ty = torch.from_numpy(ys).to(dev)
    taus = torch.from_numpy(np.random.RandomState(1).uniform(1, 3, size=C)).to(dev)
    fn = lambda: _native.poly_gauss_grad(q0, A, ty, taus)
    import time
    t = time.perf_counter()
    while time.perf_counter() - t < 0.2:
        fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        g = fn()
    e1.record(); torch.cuda.synchronize()
    out['%dx%dx%d' % (C, K, N)] = {'us': e0.elapsed_time(e1) / 40 * 1e3,
                                   'sha': hashlib.sha1(g.cpu().numpy().tobytes()).hexdigest()[:12]}
print(json.dumps(out))

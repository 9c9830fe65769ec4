#!/usr/bin/env python3
"""Row reductions in np.sum order: time per launch and HBM rate at a few shapes, the
polynomial chi^2 and the pair-distance log-prob (element functions with loads / gathers)."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
from binf_amd.example.distance import make_distance_likelihood
dev = torch.device('cuda:0')


def timed(fn, n=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


out = {}
for C, D in ((4096, 1024), (4096, 1000), (8192, 33), (4096, 16384), (65536, 128)):
    p = torch.randn((C, D), dtype=torch.float64, device=dev)
    t = timed(lambda: _native.row_sum(p, _native.ROW_SUMSQ, scale=0.5))
    out['row_sumsq_%dx%d' % (C, D)] = {'us': t * 1e6, 'TBps': 8.0 * C * D / t / 1e12}
for C, K, N in ((8192, 33, 16384), (4096, 4, 20)):
    xs = torch.linspace(-1, 1, N, dtype=torch.float64, device=dev)
    ys = torch.randn(N, dtype=torch.float64, device=dev)
    th = torch.randn((C, K), dtype=torch.float64, device=dev)
    t = timed(lambda: _native.poly_gauss_logp(th, xs, ys, 2.5), 30, 5)
    out['poly_logp_%dx%dx%d' % (C, K, N)] = {'us': t * 1e6}
for C in (256, 2048):
    n = 256
    rs = np.random.RandomState(0)
    truth = rs.standard_normal((n, 3)) * 2.0
    I, J = np.triu_indices(n, 1)
    d = np.sqrt(np.sum((truth[I] - truth[J]) ** 2, axis=1))
    lik = make_distance_likelihood(np.abs(d + 0.05 * rs.standard_normal(len(d))), n)
    x = torch.from_numpy(truth.reshape(-1)[None, :] + 0.1 * rs.standard_normal((C, 3 * n))).to(dev)
    t = timed(lambda: lik.log_prob(coordinates=x, precision=4.0), 50, 5)
    out['pairdist_logp_%d_chains' % C] = {'us': t * 1e6}
print(json.dumps(out))

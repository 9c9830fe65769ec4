#!/usr/bin/env python3
"""cProfile of GibbsSampler.sample() as one launch (host side): where the ~50 us go."""
import cProfile, pstats, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_gibbs_n import build
dev = torch.device('cuda:0')
for move in ('hmc', 'rwmc'):
    g = build(4096, move, dev)
    for _ in range(50):
        g.sample()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(2000):
        g.sample()
    dt = (time.perf_counter() - t) / 2000
    torch.cuda.synchronize()
    print(move, 'host us per sweep: %.1f' % (dt * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        g.sample()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14)
    print(s.getvalue()[:3200])

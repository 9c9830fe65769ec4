"""Timing of the posterior-predictive grid kernel (development aid):
python scripts/probe_predict.py  -> us per launch for the example's grid and a large one."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
dev = torch.device('cuda:0')
out = {}
for S, nx, ny in ((500, 100, 150), (51200, 100, 150), (4096, 1000, 1000)):
    mock = torch.randn((S, nx), dtype=torch.float64, device=dev)
    tau = torch.rand(S, dtype=torch.float64, device=dev) + 1.0
    ys = torch.randn((nx, ny), dtype=torch.float64, device=dev)
    h = 0.5 * np.log(2 * np.pi)
    for _ in range(3):
        _native.predictive_density(mock, tau, ys, h)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        _native.predictive_density(mock, tau, ys, h)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    out['S=%d grid=%dx%d' % (S, nx, ny)] = {'us': us, 'terms_per_s': S * nx * ny / us * 1e6}
print(json.dumps(out))

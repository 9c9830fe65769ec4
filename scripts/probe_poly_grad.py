"""Gradient kernel alone at C3 / C4's per-GPU shapes (development aid): time per
launch and useful TFLOP/s for the environment's BINF_POLY_GRAD_* settings."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
from binf_amd.example.likelihood import POLYVAL, ForwardModel
dev = torch.device('cuda:0')
K, N = 33, 16384
xs = np.linspace(-1, 1, N); ys = np.random.RandomState(9).standard_normal(N)
fwm = ForwardModel(xs, POLYVAL); A = fwm.design_matrix(K, dev); ty = torch.from_numpy(ys).to(dev)
out = {}
import time
for C in (8192, 4096, 2048, 1024):
    q0 = torch.from_numpy(np.random.RandomState(8).standard_normal((C, K))).to(dev)
    t_s = time.perf_counter()
    while time.perf_counter() - t_s < 0.25:          # settle: the clock controller needs ~50 ms of load
        _native.poly_gauss_grad(q0, A, ty, 2.5); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): _native.poly_gauss_grad(q0, A, ty, 2.5)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 40
    out[C] = {'ms': round(t * 1e3, 4), 'TFLOPs': round(4.0 * K * N * C / t / 1e12, 2)}
print(os.environ.get('BINF_POLY_GRAD_CT'), os.environ.get('BINF_POLY_GRAD_WGS'), json.dumps(out))

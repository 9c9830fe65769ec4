"""Fused small-data polynomial transition (binf_hmc_sample_poly_f64) alone:
example_script.py's shape (K = 4, 20 data points, 50 leapfrog steps), device
time per launch vs chain count (development aid)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
dev = torch.device('cuda:0')
K, N, L = 4, 20, 50
rs = np.random.RandomState(0)
xs = torch.from_numpy(np.linspace(-2, 2, N)).to(dev)
ys = torch.from_numpy(rs.standard_normal(N)).to(dev)
means = torch.zeros(K, dtype=torch.float64, device=dev)
var = torch.full((K,), 5.0, dtype=torch.float64, device=dev)
out = {}
for C in (4096, 65536, 1 << 20):
    q0 = torch.ones((C, K), dtype=torch.float64, device=dev)
    p0 = torch.randn((C, K), dtype=torch.float64, device=dev)
    u = torch.rand(C, dtype=torch.float64, device=dev)
    tau = torch.full((C,), 2.5, dtype=torch.float64, device=dev)
    qo = torch.empty_like(q0)
    acc = torch.empty(C, dtype=torch.uint8, device=dev)
    f = lambda: _native.hmc_sample_poly(q0, p0, u, qo, acc, None, None, None, xs, ys, tau, means, var,
                                        True, None, None, 0.02, None, L, False, 1.05, 0.95)
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    flops = C * (L + 1) * N * 13.0          # FP64 lane-operations of the force loop
    out[C] = {'us': dt * 1e6, 'chain_steps_per_s': C * L / dt, 'T_lane_ops_per_s': flops / dt / 1e12,
              'acceptance': float(acc.double().mean())}
print(json.dumps(out))

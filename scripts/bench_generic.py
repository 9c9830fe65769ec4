"""Per-step (generic) tier on a user-style PDF written with torch ops, at C2's
shape: what a plug-in posterior without a fused kernel costs (development aid)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')


class TorchGaussian(object):
    def __init__(self, k, x0):
        self.k, self.x0 = k, x0

    def log_prob(self, x):
        return (-0.5 * self.k) * _native.row_sum(x, _native.ROW_SUMSQ_SHIFT, shift=self.x0)

    def gradient(self, x):
        return self.k * (x - self.x0)


out = {}
for C, D, L, graph in ((4096, 1024, 20, False), (256, 768, 20, False), (256, 768, 20, True),
                       (2048, 64, 10, False), (2048, 64, 10, True), (64, 20000, 10, False)):
    q0 = torch.randn((C, D), dtype=torch.float64, device=dev)
    s = HMCSampler(TorchGaussian(1.0, 0.0), q0, 0.05, L, variable_name='x', rng=DeviceRNG(0, dev), graph=graph)
    for _ in range(3): s.sample()
    torch.cuda.synchronize(); t = time.perf_counter()
    K = 10
    for _ in range(K): s.sample()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / K
    bytes_step = 32.0 * D * C          # unfused kick + drift traffic per leapfrog step
    out['%dx%d L=%d%s' % (C, D, L, ' graph' if graph else '')] = {'ms_per_sample': dt * 1e3, 'us_per_leapfrog_step': dt / L * 1e6,
                                     'chain_steps_per_s': C * L / dt,
                                     'kick_drift_GBps_if_alone': bytes_step * L / dt / 1e9}
print(json.dumps(out))

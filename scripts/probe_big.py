#!/usr/bin/env python3
"""Long chains (csrc/hmc_gauss_big.hip): time per sample() and, under rocprofv3
--kernel-trace --stats, the share of its kernels.  4096 chains x 16384 dims, L = 20."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')
C, D, L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 16384, 20
out = {}
for name, rng in (('supplied', None), ('device_rng', DeviceRNG(0, dev))):
    q0 = torch.randn((C, D), dtype=torch.float64, device=dev)
    s = HMCSampler(IsotropicGaussian(), q0, 0.01, L, variable_name='x', **({'rng': rng} if rng else {}))
    p0 = torch.randn((C, D), dtype=torch.float64, device=dev)
    u = torch.rand(C, dtype=torch.float64, device=dev)
    fn = (lambda: s.sample(p0=p0, u=u)) if rng is None else s.sample
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 30
    out[name] = {'ms_per_sample': t * 1e3, 'algorithmic_TBps': (24.0 * D + 25) * C / t / 1e12,
                 'acceptance': float(s.acceptance_rate.mean())}
    del s, q0, p0
    torch.cuda.empty_cache()
print(json.dumps(out))

# n transitions from one call (binf_hmc_sample_n_gauss_big_*), every state recorded,
# against the loop of sample() calls + stacking the states
n = 8
for name, mk in (('loop_of_sample_calls', False), ('one_call_sample_n', True)):
    q0 = torch.randn((C, D), dtype=torch.float64, device=dev)
    s = HMCSampler(IsotropicGaussian(), q0, 0.01, L, variable_name='x', rng=DeviceRNG(0, dev))
    buf = torch.empty((n, C, D), dtype=torch.float64, device=dev)

    def run():
        if mk:
            s.sample_n(n, out=buf)
        else:
            for i in range(n):
                buf[i].copy_(s.sample())
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / (5 * n)
    out[name] = {'ms_per_recorded_transition': t * 1e3, 'algorithmic_TBps': (24.0 * D + 25) * C / t / 1e12}
    del s, q0, buf
    torch.cuda.empty_cache()
print(json.dumps(out))

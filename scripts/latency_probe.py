"""Wall time of one HMCSampler.sample() call for small batches (development aid):
the reference's own use is ONE chain per sampler (~220 us per sample() on a
CPU core at D = 1024, L = 20)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')
for C, D in ((1, 4), (1, 1024), (64, 1024), (4096, 1024)):
    for name, rng in (('host np.random', None), ('device rng', DeviceRNG(0, dev))):
        q0 = torch.zeros((C, D), dtype=torch.float64, device=dev) if C > 1 else torch.zeros(D, dtype=torch.float64, device=dev)
        s = HMCSampler(IsotropicGaussian(), q0, 0.05, 20, variable_name='x', rng=rng)
        for _ in range(20): s.sample()
        torch.cuda.synchronize(); t = time.perf_counter()
        K = 200 if C * D < 1e6 else 30
        for _ in range(K): s.sample()
        torch.cuda.synchronize()
        print('C=%d D=%d %-15s %.1f us per sample()' % (C, D, name, (time.perf_counter() - t) / K * 1e6))

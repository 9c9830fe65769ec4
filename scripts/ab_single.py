"""A/B: one transition per launch through the single-transition kernel vs the
persistent kernel with n = 1 (development aid)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native as N
dev = torch.device('cuda:0')
C, D, L, K, P = 4096, 1024, 20, 400, 16
gen = torch.Generator(device=dev); gen.manual_seed(0)
q = [torch.randn((C, D), dtype=torch.float64, device=dev, generator=gen) for _ in range(2)]
pool = [torch.randn((C, D), dtype=torch.float64, device=dev, generator=gen) for _ in range(P)]
u = torch.rand(C, dtype=torch.float64, device=dev, generator=gen)
acc = torch.empty(C, dtype=torch.uint8, device=dev)
nacc = torch.zeros(C, dtype=torch.int64, device=dev)
for mode in (0, 1):
    for name in ('single', 'persist(n=1)'):
        for rep in range(2):
            torch.cuda.synchronize(); t = time.perf_counter()
            for i in range(K):
                src, dst = q[i & 1], q[1 - (i & 1)]
                if name == 'single':
                    N.hmc_sample_gauss(src, pool[i % P], u, dst, acc, nacc, None, None, 0.05, None, L, 1.0, 0.0, False, 1.05, 0.95, mode)
                else:
                    N.hmc_sample_n_gauss(src, pool[i % P], u, dst, None, acc, nacc, None, None, 0.05, None, L, 1, 1, 1.0, 0.0, 0, 1.05, 0.95, mode)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / K
        print('%s mode=%d: %.2f us/launch' % (name, mode, dt * 1e6))

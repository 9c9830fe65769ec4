"""Few chains (strong-scaling shares) with device draws: generator fused into the
one-wave-per-chain kernel vs stand-alone generator kernels + chains split over waves."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')
D, L, F = 1024, 20, 64
for C in (256, 512, 1024, 2048, 4096):
    rec = torch.empty((F, C, D), dtype=torch.float64, device=dev)
    row = []
    for fused in (True, False):
        s = HMCSampler(IsotropicGaussian(), torch.randn((C, D), dtype=torch.float64, device=dev), 0.05, L,
                       variable_name='x', rng=DeviceRNG(0, dev, fused=fused))
        t_s = time.perf_counter()
        while time.perf_counter() - t_s < 0.15:
            s.sample_n(F, out=rec); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): s.sample_n(F, out=rec)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / (10 * F)
        row.append('%s %.2f us (%.2e)' % ('fused' if fused else 'separate', t * 1e6, C * L / t))
    print('C=%d' % C, ' | '.join(row))

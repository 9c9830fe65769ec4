"""Sustained rate of the headline kernel over seconds (does the settled rate hold?)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')
C, D, L, F = 4096, 1024, 20, 64
gen = torch.Generator(device=dev); gen.manual_seed(1)
p = [torch.randn((F, C, D), dtype=torch.float64, device=dev, generator=gen) for _ in range(3)]
u = [torch.rand((F, C), dtype=torch.float64, device=dev, generator=gen) for _ in range(3)]
rec = [torch.empty((F, C, D), dtype=torch.float64, device=dev) for _ in range(2)]
for name in ('exact', 'fma', 'exact_rng'):
    mode = 'fma' if name == 'fma' else 'exact'
    s = HMCSampler(IsotropicGaussian(), torch.randn((C, D), dtype=torch.float64, device=dev), 0.05, L,
                   variable_name='x', mode=mode, rng=DeviceRNG(0, dev) if name == 'exact_rng' else None)
    out = []
    for blk in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(250):
            if name == 'exact_rng': s.sample_n(F, out=rec[i % 2])
            else: s.sample_n(F, p0=p[i % 3], u=u[i % 3], out=rec[i % 2])
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / (250 * F))
    print(name, 'us/transition per block of 250 launches:', ' '.join('%.2f' % x for x in out))
    time.sleep(1.0)

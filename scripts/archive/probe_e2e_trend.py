import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')
C, D, L, F = 4096, 1024, 20, 64
buf = torch.empty((F, C, D), dtype=torch.float64, device=dev)
for q0kind in ('zeros', 'randn'):
    q0 = torch.zeros((C, D), dtype=torch.float64, device=dev) if q0kind == 'zeros' else torch.randn((C, D), dtype=torch.float64, device=dev)
    s = HMCSampler(IsotropicGaussian(), q0, 0.05, L, variable_name='x', rng=DeviceRNG(0, dev))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
    ev[0].record()
    for i in range(30):
        s.sample_n(F, out=buf)
        ev[i + 1].record()
    torch.cuda.synchronize()
    print(q0kind, ' '.join('%.1f' % (ev[i].elapsed_time(ev[i + 1]) * 1e3 / F) for i in range(30)), 'acc %.3f' % float(s.acceptance_rate.mean()))

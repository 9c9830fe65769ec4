"""Per-launch time of the headline kernel over a long run (clock / power settling)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
dev = torch.device('cuda:0')
C, D, L, F = 4096, 1024, 20, 64
gen = torch.Generator(device=dev); gen.manual_seed(1)
p = [torch.randn((F, C, D), dtype=torch.float64, device=dev, generator=gen) for _ in range(3)]
u = [torch.rand((F, C), dtype=torch.float64, device=dev, generator=gen) for _ in range(3)]
rec = [torch.empty((F, C, D), dtype=torch.float64, device=dev) for _ in range(2)]
for mode in ('exact', 'fma'):
    s = HMCSampler(IsotropicGaussian(), torch.randn((C, D), dtype=torch.float64, device=dev), 0.05, L, variable_name='x', mode=mode)
    N = 160
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    ev[0].record()
    for i in range(N):
        s.sample_n(F, p0=p[i % 3], u=u[i % 3], out=rec[i % 2])
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = [ev[i].elapsed_time(ev[i + 1]) * 1e3 / F for i in range(N)]
    print(mode, ' '.join('%.1f' % x for x in t[:40]))
    print(mode, 'mean of launches 0-19 %.2f  20-39 %.2f  40-79 %.2f  80-159 %.2f' % (sum(t[:20]) / 20, sum(t[20:40]) / 20, sum(t[40:80]) / 40, sum(t[80:]) / 80))
    import time; time.sleep(2.0)

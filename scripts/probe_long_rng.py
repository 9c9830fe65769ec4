"""Long chains (4096 x 16384) with device draws: generated in the chunked kernels vs
the stand-alone generator kernels; and with draws supplied (development aid)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')
C, D, L = 4096, 16384, 20
q0 = torch.randn((C, D), dtype=torch.float64, device=dev)
p0 = torch.randn((C, D), dtype=torch.float64, device=dev)
u = torch.rand(C, dtype=torch.float64, device=dev)
def timed(fn, n=12, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, rng, kw in (('draws supplied', None, dict(p0=p0, u=u)), ('in-kernel draws', DeviceRNG(0, dev), {}),
                      ('stand-alone generator', DeviceRNG(0, dev, fused=False), {})):
    s = HMCSampler(IsotropicGaussian(), q0, 0.01, L, variable_name='x', rng=rng)
    t = timed(lambda: s.sample(**kw))
    print('%-22s %.3f ms per sample  %.2e chain-steps/s' % (name, t, C * L / (t * 1e-3)))

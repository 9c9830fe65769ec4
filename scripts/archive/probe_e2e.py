"""C2 with in-kernel draws, longer timing loop (development aid for A/B of builds:
BINF_LIB_OVERRIDE=<other .so>)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')
C, D, L, F = 4096, 1024, 20, 64
buf = torch.empty((F, C, D), dtype=torch.float64, device=dev)
res = []
for mode in ('exact', 'fma'):
    s = HMCSampler(IsotropicGaussian(), torch.zeros((C, D), dtype=torch.float64, device=dev), 0.05, L,
                   variable_name='x', rng=DeviceRNG(0, dev), mode=mode)
    for _ in range(4): s.sample_n(F, out=buf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(24): s.sample_n(F, out=buf)
    e1.record(); torch.cuda.synchronize()
    res.append('%s %.2f us' % (mode, e0.elapsed_time(e1) * 1e3 / (24 * F)))
print(os.environ.get('BINF_LIB_OVERRIDE', 'HEAD'), ' '.join(res))

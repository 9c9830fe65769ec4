"""End-to-end throughput mode: HMCSampler.sample_n with DeviceRNG (draws
generated on the device inside the timed region), C2 shape."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')
C, D, L, F = 4096, 1024, 20, 64
out = {}
buf = torch.empty((F, C, D), dtype=torch.float64, device=dev)
for kind in ('normal', 'uniform'):
    _native.rng_fill(kind, buf, 1, 0); torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(5): _native.rng_fill(kind, buf, 1, i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    out['rng_%s_Gdraws_per_s' % kind] = buf.numel() / dt / 1e9
    out['rng_%s_us_per_4M' % kind] = dt / F * 1e6
for mode in ('exact', 'fma'):
    s = HMCSampler(IsotropicGaussian(), torch.zeros((C, D), dtype=torch.float64, device=dev),
                   0.05, L, variable_name='x', rng=DeviceRNG(0, dev), mode=mode)
    s.sample_n(F); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(8): s.sample_n(F)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / (8 * F)
    out['e2e_%s_us_per_transition' % mode] = dt * 1e6
    out['e2e_%s_chain_steps_per_s' % mode] = C * L / dt
print(json.dumps(out))

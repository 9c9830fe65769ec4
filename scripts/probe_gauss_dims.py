"""Fused Gaussian HMC across chain lengths (development aid): element-steps per second by D,
to spot cliffs between the kernel layouts (several chains per wave, one wave per chain, ragged
trees, 2 / 4 / 8 waves per chain, the chunked long-chain path)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')
out = {}
L, n = 20, 16
DIMS = (8, 33, 64, 100, 128, 200, 256, 300, 512, 700, 768, 1000, 1024, 1025, 1500, 2048, 3000, 4096, 5000, 8192, 9000, 16384, 40000)
HOST = bool(os.environ.get('PROBE_HOSTDRAWS'))      # draws supplied from HBM instead of made in the kernel
if HOST:
    DIMS = (33, 100, 128, 200, 300, 700, 768, 1000, 1024)
for D in DIMS:
    # whole rounds of waves on 1024 SIMDs x 4 waves: chains per wave = 64 >> (3 + H) for D <= 1024
    C = int(os.environ.get('PROBE_CHAINS', 4096)) * (max(1, 1024 // max(D, 128)) if D <= 1024 else 1)
    if D > 1024: C = max(64, C * 1024 // (1 << (D - 1).bit_length()))
    q0 = torch.randn((C, D), dtype=torch.float64, device=dev)
    s = HMCSampler(IsotropicGaussian(), q0, 0.05 * (1024.0 / D) ** 0.25, L, variable_name='x', rng=DeviceRNG(0, dev))
    kw = {}
    if HOST:
        kw = dict(p0=torch.randn((n, C, D), dtype=torch.float64, device=dev), u=torch.rand((n, C), dtype=torch.float64, device=dev))
    for _ in range(2): s.sample_n(n, record=False, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): s.sample_n(n, record=False, **kw)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / (4 * n)
    out[D] = {'chains': C, 'us_per_transition': t * 1e6, 'element_steps_per_s': C * D * L / t,
              'acceptance': float(s.acceptance_rate.mean())}
    print(D, out[D], flush=True)

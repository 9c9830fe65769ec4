#!/usr/bin/env python3
"""Host cost of one C-ABI launch through binf_amd/_native.py (what bounds small batches):
wall time per call without waiting for the GPU, and a cProfile of the wrapper."""
import cProfile, pstats, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
dev = torch.device('cuda:0')
C, D = 256, 768
p = torch.randn((C, D), dtype=torch.float64, device=dev)
lp = torch.randn(C, dtype=torch.float64, device=dev)
g = torch.randn((C, D), dtype=torch.float64, device=dev)
q = torch.randn((C, D), dtype=torch.float64, device=dev)


def bench(fn, n=3000):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    return dt * 1e6


print('hmc_energy          %.1f us' % bench(lambda: _native.hmc_energy(p, lp)))
print('leapfrog_kick_drift %.1f us' % bench(lambda: _native.leapfrog_kick_drift(q, p, g, 1e-9)))
print('sum_terms(3)        %.1f us' % bench(lambda: _native.sum_terms([lp, 0.5, lp])))
print('torch.empty         %.1f us' % bench(lambda: torch.empty(C, dtype=torch.float64, device=dev)))
print('stream_handle       %.2f us' % bench(lambda: _native.stream_handle(dev)))
print('dptr                %.2f us' % bench(lambda: _native.dptr(p, numel=C * D, name='p')))
L = _native.lib()
st = _native.stream_handle(dev)
out = torch.empty(C, dtype=torch.float64, device=dev)
print('raw ctypes call     %.1f us' % bench(lambda: L.binf_hmc_energy_f64(p.data_ptr(), lp.data_ptr(), out.data_ptr(), C, D, st)))
pr = cProfile.Profile()
pr.enable()
for _ in range(3000):
    _native.hmc_energy(p, lp)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(8)
print(s.getvalue()[:1800])

"""Where one HMCSampler.sample() call (one transition per launch) spends its time
at C2's shape: host cost of a call (tiny batch, draws supplied), device time of
the launch (HIP events around back-to-back launches) -- development aid for the
single-call path GibbsSampler drives (binf/samplers/gibbs.py:148)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
dev = torch.device('cuda:0')
out = {}
D, L = 1024, 20
for C in (64, 4096):
    q0 = torch.randn((C, D), dtype=torch.float64, device=dev)
    P = 16
    p = [torch.randn((C, D), dtype=torch.float64, device=dev) for _ in range(P)]
    u = [torch.rand(C, dtype=torch.float64, device=dev) for _ in range(P)]
    s = HMCSampler(IsotropicGaussian(), q0, 0.05, L, variable_name='x')
    for i in range(50): s.sample(p0=p[i % P], u=u[i % P])
    torch.cuda.synchronize()
    K = 1000
    t = time.perf_counter()
    for i in range(K): s.sample(p0=p[i % P], u=u[i % P])
    t_issue = (time.perf_counter() - t) / K
    torch.cuda.synchronize()
    t_total = (time.perf_counter() - t) / K
    out['C=%d' % C] = {'host_issue_us_per_call': t_issue * 1e6, 'wall_us_per_call': t_total * 1e6}
    # the raw C-ABI call alone, same buffers every time (no Python sampler logic)
    q_out = torch.empty_like(q0); acc = torch.empty(C, dtype=torch.uint8, device=dev)
    nacc = torch.zeros(C, dtype=torch.int64, device=dev)
    def raw(i):
        _native.hmc_sample_gauss(q0, p[i % P], u[i % P], q_out, acc, nacc, None, None, 0.05, None,
                                 L, 1.0, 0.0, False, 1.05, 0.95)
    for i in range(50): raw(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(K): raw(i)
    t_issue = (time.perf_counter() - t) / K
    torch.cuda.synchronize()
    t_total = (time.perf_counter() - t) / K
    out['C=%d' % C].update({'raw_abi_issue_us': t_issue * 1e6, 'raw_abi_wall_us': t_total * 1e6})
print(json.dumps(out))

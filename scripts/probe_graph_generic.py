"""Per-step tier with a user's torch PDF: eager launches vs one HIP-graph replay of the whole
transition (development aid for HMCSampler(graph=True)).  python scripts/probe_graph_generic.py"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.samplers.hmc import HMCSampler
dev = torch.device('cuda:0')


class DoubleWell(object):
    def __init__(self, a=2.0):
        self.a = a

    def log_prob(self, x):
        w = x * x - 1.0
        return (-self.a) * (w * w).sum(dim=1)

    def gradient(self, x):
        return (4.0 * self.a) * x * (x * x - 1.0)


out = {}
for C, D, L in ((64, 64, 10), (2048, 64, 10), (4096, 1024, 20), (256, 768, 20)):
    q0 = torch.randn((C, D), dtype=torch.float64, device=dev) * 0.5 + 1
    p = torch.randn((C, D), dtype=torch.float64, device=dev)
    u = torch.rand(C, dtype=torch.float64, device=dev)
    s = HMCSampler(DoubleWell(), q0.clone(), 0.05, L, variable_name='x')
    for _ in range(3):
        s.sample(p0=p, u=u)
    torch.cuda.synchronize()
    K = 20
    t = time.perf_counter()
    for _ in range(K):
        s.sample(p0=p, u=u)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t) / K
    # the same transition as one graph
    q_in, p_in, u_in = q0.clone(), p.clone(), u.clone()
    acc = torch.empty(C, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        s._sample_generic('x', q_in, q_in, p_in, True, u_in, acc, False)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        q_out = s._sample_generic('x', q_in, q_in, p_in, True, u_in, acc, False)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(K):
        p_in.copy_(p); q_in.copy_(q0); u_in.copy_(u)
        g.replay()
        x = q_out.clone()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t) / K
    # same result?
    s2 = HMCSampler(DoubleWell(), q0.clone(), 0.05, L, variable_name='x')
    want = s2.sample(p0=p.clone(), u=u.clone())
    out['%dx%d L=%d' % (C, D, L)] = {'eager_ms': eager * 1e3, 'graph_ms': graph * 1e3, 'speedup': eager / graph,
                                     'same_bits': bool(torch.equal(want, x))}
print(json.dumps(out, indent=1))

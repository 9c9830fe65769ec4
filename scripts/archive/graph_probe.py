"""Probe: capture one generic-tier leapfrog step in a HIP graph (torch.cuda.graph)
and replay it, vs the plain Python loop (development aid)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
from binf_amd.example.misc import make_posterior
dev = torch.device('cuda:0')
C, L, dt = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 50, 0.02
np.random.seed(0)
xs = np.linspace(-2, 2, 20)
ys = np.random.normal(np.polynomial.polynomial.polyval(xs, [2., -4., 1., 1.5]), 0.6)
post = make_posterior(xs, ys, np.polynomial.polynomial.polyval)
cond = post.conditional_factory(precision=torch.ones(C, dtype=torch.float64, device=dev))
q = torch.ones((C, 4), dtype=torch.float64, device=dev)
p = torch.randn((C, 4), dtype=torch.float64, device=dev)
grad = lambda x: cond.gradient(coefficients=x).contiguous()

def step():
    _native.leapfrog_drift(q, p, dt, None)
    _native.leapfrog_kick(p, grad(q), dt, None)

def plain(n):
    for _ in range(n):
        step()

for _ in range(3):
    plain(L)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20):
    plain(L)
torch.cuda.synchronize(); print('plain loop: %.1f us per step' % ((time.perf_counter() - t) / 20 / L * 1e6))

q0, p0 = q.clone(), p.clone()
plain(L); torch.cuda.synchronize(); qa, pa = q.clone(), p.clone()
q.copy_(q0); p.copy_(p0)
def graphed(n):
    step()                                  # warm + step 1
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(n - 1):
        g.replay()
graphed(L); torch.cuda.synchronize()
print('graph result identical:', torch.equal(q, qa), torch.equal(p, pa))
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20):
    graphed(L)
torch.cuda.synchronize(); print('capture per call + %d replays: %.1f us per step, %.1f us per call' % (L - 1, (time.perf_counter() - t) / 20 / L * 1e6, (time.perf_counter() - t) / 20 * 1e6))
t = time.perf_counter()
for _ in range(20):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
torch.cuda.synchronize(); print('capture + instantiate alone: %.1f us' % ((time.perf_counter() - t) / 20 * 1e6))

#!/usr/bin/env python3
"""cProfile of one HMCSampler.sample() on the C5 posterior (256 chains x 256 beads): the
host side of its ~15 launches."""
import cProfile, pstats, io, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.example.distance import make_distance_likelihood
from binf_amd.pdf import IsotropicGaussian
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
L = 20
rs = np.random.RandomState(0)
truth = rs.standard_normal((n, 3)) * 2.0
I, J = np.triu_indices(n, 1)
d = np.sqrt(np.sum((truth[I] - truth[J]) ** 2, axis=1))
ys = np.abs(d + 0.05 * rs.standard_normal(len(d)))
x = torch.from_numpy(truth.reshape(-1)[None, :] + 0.1 * rs.standard_normal((C, 3 * n))).to(dev)
lik = make_distance_likelihood(ys, n)
prior = IsotropicGaussian(0.05, 0.0, name='coordinates_prior', variable_name='coordinates')
cond = Posterior({lik.name: lik}, {prior.name: prior}).conditional_factory(precision=4.0)
s = HMCSampler(cond, x, 0.002, L, variable_name='coordinates', rng=DeviceRNG(0, dev))
for _ in range(20):
    s.sample()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(300):
    s.sample()
host = (time.perf_counter() - t) / 300
torch.cuda.synchronize()
total = (time.perf_counter() - t) / 300
print('host issue %.1f us per sample(), with the GPU drained %.1f us' % (host * 1e6, total * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    s.sample()
pr.disable()
torch.cuda.synchronize()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats('tottime').print_stats(22)
print(st.getvalue()[:4200])

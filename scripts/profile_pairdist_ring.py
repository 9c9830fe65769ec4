"""Kernel trace target for the pair-distance model beyond 256 beads (development aid):
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 <repo>/scripts/profile_pairdist_ring.py
20 sample() calls (L = 20) each at 1024 beads x 1024 chains (ring kernels), 1024 x 32 (a wave per
tile + chi^2 by chunks) and 4096 x 8 (tiles beyond 1024 beads)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.example.distance import make_distance_likelihood
from binf_amd.pdf import IsotropicGaussian
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')
for n, C in ((1024, 1024), (1024, 32), (4096, 8)):
    rs = np.random.RandomState(0)
    truth = rs.standard_normal((n, 3)) * 2.0
    iu = np.triu_indices(n, 1)
    ys = np.abs(np.sqrt(((truth[iu[0]] - truth[iu[1]]) ** 2).sum(-1)) + 0.05 * rs.standard_normal(len(iu[0])))
    lik = make_distance_likelihood(ys, n)
    prior = IsotropicGaussian(0.05, 0.0, name='coordinates_prior', variable_name='coordinates')
    cond = Posterior({lik.name: lik}, {prior.name: prior}).conditional_factory(precision=4.0)
    x = torch.from_numpy(truth.reshape(-1)[None, :] + 0.1 * rs.standard_normal((C, 3 * n))).to(dev)
    s = HMCSampler(cond, x, 0.001 if n <= 1024 else 0.0005, 20, variable_name='coordinates', rng=DeviceRNG(0, dev))
    for _ in range(20):
        s.sample()
    torch.cuda.synchronize()

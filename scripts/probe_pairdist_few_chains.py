"""Pair-distance sample() with FEW chains (development aid): a workgroup per chain (ring
kernels, BINF_PD_TILES=0) against a wave per tile (BINF_PD_TILES=1); default = the library's
own choice.  python scripts/probe_pairdist_few_chains.py"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.example.distance import make_distance_likelihood
from binf_amd.pdf import IsotropicGaussian
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')
out = {}
for n in (320, 512, 1000, 1024):
    rs = np.random.RandomState(0)
    truth = rs.standard_normal((n, 3)) * 2.0
    d = np.sqrt(((truth[:, None, :] - truth[None, :, :]) ** 2).sum(-1))
    iu = np.triu_indices(n, 1)
    ys = np.abs(d[iu] + 0.05 * rs.standard_normal(len(iu[0])))
    lik = make_distance_likelihood(ys, n)
    prior = IsotropicGaussian(0.05, 0.0, name='coordinates_prior', variable_name='coordinates')
    cond = Posterior({lik.name: lik}, {prior.name: prior}).conditional_factory(precision=4.0)
    for C in (1, 8, 32, 64, 128, 256, 512):
        x = torch.from_numpy(truth.reshape(-1)[None, :] + 0.1 * rs.standard_normal((C, 3 * n))).to(dev)
        s = HMCSampler(cond, x, 0.001, 20, variable_name='coordinates', rng=DeviceRNG(0, dev))
        for _ in range(3): s.sample()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): s.sample()
        torch.cuda.synchronize()
        out['n=%d C=%d' % (n, C)] = round((time.perf_counter() - t) / 10 * 1e3, 3)
print(json.dumps(out))

"""Force kernel and a whole sample() of the pair-distance model vs the number of beads
(development aid): unordered pairs per second, 256 ... 1024 beads."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
from binf_amd.example.distance import make_distance_likelihood
from binf_amd.pdf import IsotropicGaussian
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')
out = {}
import os
SHAPES = ((256, 256), (256, 2048), (384, 256), (512, 256), (512, 1024), (1024, 256), (1024, 1024))
if os.environ.get('PROBE_BIG'):
    SHAPES = ((1500, 64), (2048, 16), (2048, 128), (4096, 8), (4096, 64), (8192, 8))
elif os.environ.get('PROBE_RAGGED'):
    SHAPES = ((500, 256), (500, 1024), (512, 1024), (1000, 256), (1000, 1024), (1024, 1024), (700, 512))
for n, C in SHAPES:
    rs = np.random.RandomState(0)
    truth = rs.standard_normal((n, 3)) * 2.0
    d = np.sqrt(((truth[:, None, :] - truth[None, :, :]) ** 2).sum(-1))
    iu = np.triu_indices(n, 1)
    ys = np.abs(d[iu] + 0.05 * rs.standard_normal(len(iu[0])))
    ymat = torch.from_numpy(np.abs(d + 0.05 * rs.standard_normal((n, n)))).to(dev)
    x = torch.from_numpy(truth.reshape(-1)[None, :] + 0.1 * rs.standard_normal((C, 3 * n))).to(dev)
    packed = _native.pairdist_pack_targets(ymat)
    for _ in range(3): _native.pairdist_gauss_grad(x, ymat, 4.0, packed=packed)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): _native.pairdist_gauss_grad(x, ymat, 4.0, packed=packed)
    e1.record(); torch.cuda.synchronize()
    tg = e0.elapsed_time(e1) * 1e-3 / 20
    lik = make_distance_likelihood(ys, n)
    prior = IsotropicGaussian(0.05, 0.0, name='coordinates_prior', variable_name='coordinates')
    cond = Posterior({lik.name: lik}, {prior.name: prior}).conditional_factory(precision=4.0)
    s = HMCSampler(cond, x, 0.001, 20, variable_name='coordinates', rng=DeviceRNG(0, dev))
    for _ in range(3): s.sample()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): s.sample()
    torch.cuda.synchronize(); ts = (time.perf_counter() - t) / 10
    pairs = n * (n - 1) / 2
    out['n=%d C=%d' % (n, C)] = {'force_us': tg * 1e6, 'force_unordered_pairs_per_s': C * pairs / tg,
                                 'sample_ms': ts * 1e3, 'sample_unordered_pairs_per_s': C * pairs * 21 / ts}
print(json.dumps(out, indent=1))

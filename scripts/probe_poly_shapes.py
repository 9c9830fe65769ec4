"""HMC on the polynomial model across (K coefficients, N data, C chains): ms per sample() and the
gradient's useful TFLOP/s (4 K N flop per chain and gradient, L + 1 gradients per sample) --
development aid to spot cliffs between the fused small-data transition, the whole-tile MFMA
gradient and the general one."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.example.likelihood import POLYVAL, make_likelihood
from binf_amd.example.priors import GammaPrior, GaussianPrior
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')
L = 10
out = {}
for K, N, C in ((4, 20, 4096), (4, 1000, 4096), (4, 16384, 4096), (16, 1024, 4096), (17, 1024, 4096), (17, 1000, 4096),
                (33, 1000, 4096), (33, 1024, 4096), (33, 16384, 256), (33, 16384, 4096), (33, 16400, 4096),
                (33, 100000, 1024), (64, 16384, 4096), (48, 4096, 4096), (8, 200, 65536), (33, 16384, 64)):
    rs = np.random.RandomState(K + N)
    xs = np.linspace(-1, 1, N)
    ys = rs.standard_normal(N)
    lik = make_likelihood(xs, ys, POLYVAL)
    post = Posterior({lik.name: lik}, {'precision_prior': GammaPrior(1.0, 0.2),
                                       'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K))})
    cond = post.conditional_factory(precision=2.0)
    q0 = torch.from_numpy(rs.standard_normal((C, K)) * 0.1).to(dev)
    s = HMCSampler(cond, q0, 1e-4, L, variable_name='coefficients', rng=DeviceRNG(0, dev))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:            # clocks settle under this load
        s.sample(); torch.cuda.synchronize()
    t = time.perf_counter()
    n = 10
    for _ in range(n): s.sample()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    fused = s._fused_spec('coefficients', K, C) is not None
    out['K=%d N=%d C=%d' % (K, N, C)] = {'ms_per_sample': round(dt * 1e3, 3), 'tier': 'fused' if fused else 'per-step',
                                         'useful_TFLOPs': round(4.0 * K * N * C * (L + 1) / dt / 1e12, 2)}
    print('K=%d N=%d C=%d' % (K, N, C), out['K=%d N=%d C=%d' % (K, N, C)], flush=True)

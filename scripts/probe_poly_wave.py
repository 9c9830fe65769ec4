"""Fused polynomial transition for medium data sets (one wave per chain) against
the per-step tier (development aid)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.example.likelihood import POLYVAL, ForwardModel, GaussianErrorModel
from binf_amd.example.priors import GammaPrior, GaussianPrior
from binf_amd.pdf.likelihoods import Likelihood
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')
def timed(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for K, N, L in ((4, 200, 50), (8, 512, 20), (16, 1024, 20)):
    xs = np.linspace(-1, 1, N); rs = np.random.RandomState(0)
    ys = POLYVAL(xs, rs.standard_normal(K)) + 0.5 * rs.standard_normal(N)
    lik = Likelihood('points', ForwardModel(xs, POLYVAL), GaussianErrorModel(ys))
    post = Posterior({lik.name: lik}, {'precision_prior': GammaPrior(1.0, 0.2),
                                       'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K) * 5)})
    cond = post.conditional_factory(precision=2.0)
    for C in (64, 1024, 8192, 65536):
        q0 = torch.from_numpy(0.1 * rs.standard_normal((C, K))).to(dev)
        row = []
        for fused in (True, False):
            s = HMCSampler(cond, q0, 1e-3 / K, L, variable_name='coefficients', rng=DeviceRNG(1, dev))
            s.fused_transition = fused
            row.append(timed(s.sample))
        print('K=%2d N=%4d L=%d C=%6d  fused %.3f ms  per-step %.3f ms  x%.1f' % (K, N, L, C, row[0], row[1], row[1] / row[0]))

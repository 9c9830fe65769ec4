import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from binf_amd.example.likelihood import POLYVAL, make_likelihood
from binf_amd.example.priors import GammaPrior, GaussianPrior
from binf_amd.pdf.posteriors import Posterior
from binf_amd.example import native_poly
dev = torch.device('cuda:0')
C, K, N = 8192, 33, 16384
xs = np.linspace(-1, 1, N)
ys = POLYVAL(xs, np.random.RandomState(7).standard_normal(K)) + np.random.RandomState(9).standard_normal(N) / np.sqrt(2.5)
q0 = torch.from_numpy(np.random.RandomState(8).standard_normal((C, K))).to(dev)
lik = make_likelihood(xs, ys, POLYVAL)
post = Posterior({lik.name: lik}, {'precision_prior': GammaPrior(1.0, 0.2), 'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K) * 5)})
cond = post.conditional_factory(precision=2.5)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print('same q, memo on :', t(lambda: cond.log_prob(coefficients=q0)), 'us')
q1 = q0.clone()
print('clone q, memo on:', t(lambda: cond.log_prob(coefficients=q1)), 'us')
native_poly.USE_CHI2_MEMO = False
print('memo off        :', t(lambda: cond.log_prob(coefficients=q0)), 'us')
native_poly.USE_CHI2_MEMO = True
qs = [q0 + 1e-3 * i for i in range(4)]
i = [0]
def alt():
    i[0] += 1
    cond.log_prob(coefficients=qs[i[0] % 4])
print('changing q, memo on:', t(alt), 'us')
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
for use in (True, False, True, False):
    native_poly.USE_CHI2_MEMO = use
    s = HMCSampler(cond, q0, 2e-4, 20, variable_name='coefficients', rng=DeviceRNG(1, dev))
    for _ in range(3): s.sample()
    print('sample(), memo', use, ':', t(s.sample, 10), 'us', float(s.acceptance_rate.mean()))

"""Secondary benchmark (not the headline): BASELINE configs C3 / C4 on one GPU.
C3: polynomial ForwardModel (degree 32 -> K=33 coefficients, N=16384 data) +
Gaussian error model, 8192 chains, HMC with L=20 through the class stack.
C4: the same inside Gibbs (HMC + conjugate precision update)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
from binf_amd.example.likelihood import POLYVAL, ForwardModel, make_likelihood
from binf_amd.example.priors import GammaPrior, GaussianPrior
from binf_amd.example.samplers import make_hmc_sampler
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers import BinfState
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG

ap = argparse.ArgumentParser()
ap.add_argument('--chains', type=int, default=8192)
ap.add_argument('--K', type=int, default=33)
ap.add_argument('--N', type=int, default=16384)
ap.add_argument('--L', type=int, default=20)
ap.add_argument('--samples', type=int, default=3)
a = ap.parse_args()
dev = torch.device('cuda:0')
C, K, N, L = a.chains, a.K, a.N, a.L
xs = np.linspace(-1, 1, N)
c_true = np.random.RandomState(7).standard_normal(K)
ys = POLYVAL(xs, c_true) + np.random.RandomState(9).standard_normal(N) / np.sqrt(2.5)
q0 = torch.from_numpy(np.random.RandomState(8).standard_normal((C, K))).to(dev)


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


fwm = ForwardModel(xs, POLYVAL)
A = fwm.design_matrix(K, dev)
tx = fwm.xs_device(dev)
ty = torch.from_numpy(ys).to(dev)
t_grad = timed(lambda: _native.poly_gauss_grad(q0, A, ty, 2.5), 20)
t_logp = timed(lambda: _native.poly_gauss_logp(q0, tx, ty, 2.5), 20)
flops = 4.0 * K * N * C
res = {'config': {'chains': C, 'K': K, 'N': N, 'L': L},
       'grad_kernel_ms': t_grad * 1e3, 'grad_TFLOPs': flops / t_grad / 1e12,
       'logp_kernel_ms': t_logp * 1e3}

lik = make_likelihood(xs, ys, POLYVAL)
post = Posterior({lik.name: lik}, {'precision_prior': GammaPrior(1.0, 0.2),
                                   'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K) * 5)})
cond = post.conditional_factory(precision=2.5)
s = HMCSampler(cond, q0, 2e-4, L, variable_name='coefficients', rng=DeviceRNG(1, dev))
t_hmc = timed(s.sample, a.samples)
res['C3_hmc_sample_ms'] = t_hmc * 1e3
res['C3_chain_leapfrog_steps_per_s'] = C * L / t_hmc
res['C3_acceptance'] = float(s.acceptance_rate.mean())

grng = DeviceRNG(5, dev)
gamma = grng.gamma


start = BinfState(dict(coefficients=q0, precision=torch.full((C,), 2.5, dtype=torch.float64, device=dev)))
gips = make_hmc_sampler(post, 2e-4, L, start, rng=DeviceRNG(2, dev), gamma=gamma)
t_gibbs = timed(gips.sample, a.samples)
res['C4_gibbs_sweep_ms'] = t_gibbs * 1e3
res['C4_chain_leapfrog_steps_per_s'] = C * L / t_gibbs
print(json.dumps(res))

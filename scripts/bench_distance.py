"""Secondary benchmark: BASELINE config C5 per-GPU share (3 x 256 coordinates,
256 chains = 2048 / 8), HMC through the class stack (generic tier around the
fused all-pairs force kernel)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.example.distance import make_distance_likelihood
from binf_amd.pdf import IsotropicGaussian
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG

dev = torch.device('cuda:0')
n, L = 256, 20
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rs = np.random.RandomState(0)
truth = rs.standard_normal((n, 3)) * 2.0
I_, J_ = np.triu_indices(n, 1)
ys = np.abs(np.sqrt(np.sum((truth[I_] - truth[J_]) ** 2, axis=1)) + 0.05 * rs.standard_normal(n * (n - 1) // 2))
x = torch.from_numpy(truth.reshape(-1)[None, :] + 0.1 * rs.standard_normal((C, 3 * n))).to(dev)
lik = make_distance_likelihood(ys, n)
prior = IsotropicGaussian(0.05, 0.0, name='coordinates_prior', variable_name='coordinates')
cond = Posterior({lik.name: lik}, {prior.name: prior}).conditional_factory(precision=4.0)


def timed(fn, k):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:        # let the clocks settle under this load
        fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / k


t_g = timed(lambda: lik.gradient(coordinates=x, precision=4.0), 20)
from binf_amd import _native
_I, _J = lik.forward_model.pair_index(dev)
_ty = lik.error_model.ys_device(dev)
t_l = timed(lambda: _native.pairdist_gauss_logp(x, _I, _J, _ty, 4.0), 20)       # every chain summed
t_lm = timed(lambda: lik.log_prob(coordinates=x, precision=4.0), 20)            # unchanged chains: memo hit
s = HMCSampler(cond, x, 0.002, L, variable_name='coordinates', rng=DeviceRNG(0, dev))
t_h = timed(s.sample, 30)
print(json.dumps({'config': {'chains': C, 'beads': n, 'L': L},
                  'force_kernel_ms': t_g * 1e3, 'pair_interactions_per_s': C * n * n / t_g,
                  'logp_ms': t_l * 1e3, 'logp_memo_hit_ms': t_lm * 1e3, 'hmc_sample_ms': t_h * 1e3,
                  'chain_leapfrog_steps_per_s': C * L / t_h,
                  'acceptance': float(s.acceptance_rate.mean())}))

#!/usr/bin/env python3
"""A user forward model WITHOUT a fused kernel of its own through full sample() calls
on the per-step tier (run under rocprofv3 --kernel-trace --stats): every kernel of the
sampling loop should be the library's -- the chain-rule contraction
(binf_jacobian_contract_f64) and the Posterior's term sums (binf_sum_terms_f64) used to
be torch.bmm and torch adds."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.example.likelihood import POLYVAL, ForwardModel, GaussianErrorModel
from binf_amd.example.priors import GammaPrior, GaussianPrior
from binf_amd.pdf.likelihoods import Likelihood
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG


class PlainPolynomial(ForwardModel):
    def _evaluate(self, coefficients):           # an override: no fused kernel may be assumed
        return ForwardModel._evaluate(self, coefficients)


def main():
    dev = torch.device('cuda:0')
    K, N, C, L = 8, 512, 4096, 20
    rs = np.random.RandomState(0)
    xs = np.linspace(-1, 1, N)
    ys = POLYVAL(xs, rs.standard_normal(K)) + rs.standard_normal(N) / np.sqrt(2.5)
    lik = Likelihood('points', PlainPolynomial(xs, POLYVAL), GaussianErrorModel(ys))
    post = Posterior({lik.name: lik}, {'precision_prior': GammaPrior(1.0, 0.2),
                                       'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K) * 5)})
    tau = torch.full((C,), 2.5, dtype=torch.float64, device=dev)
    cond = post.conditional_factory(precision=tau)
    assert cond.native_hmc_spec('coefficients') is None
    q0 = torch.from_numpy(rs.standard_normal((C, K))).to(dev)
    s = HMCSampler(cond, q0, 1e-3, L, variable_name='coefficients', rng=DeviceRNG(0, dev, fused=False))
    torch.cuda.synchronize()
    for _ in range(20):
        s.sample()
    torch.cuda.synchronize()
    print('acceptance %.3f' % float(s.acceptance_rate.mean()))


if __name__ == '__main__':
    main()

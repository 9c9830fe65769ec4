#!/usr/bin/env python3
"""
A user-written PDF on the per-step tier: anything with ``log_prob(**vars) ->
[C]`` and ``gradient(**vars) -> [C x D]`` written with torch ops plugs into
``HMCSampler`` the way a duck-typed pdf plugs into the reference's sampler
(``binf/samplers/hmc.py:34-52`` only ever calls ``pdf.log_prob`` and
``pdf.gradient``).  The momentum draw, the leapfrog kick-drift, the energy and
the accept/select run in the library's HIP kernels; the pdf's own two methods
run as whatever torch launches they are made of.

Target here: independent double wells, log p(x) = -a * sum_d (x_d^2 - 1)^2,
which no built-in pdf covers.  Each coordinate should end up near +1 or -1 and
hop between the two.

  python examples/custom_pdf.py --chains 2048 --dims 64 --draws 400
  python examples/custom_pdf.py --graph      # the transition's launches replayed from one HIP graph
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG


class DoubleWell(object):
    """log p(x) = -a * sum_d (x_d^2 - 1)^2 per chain (row of ``x``)."""

    variables = {'x'}

    def __init__(self, a=2.0):
        self.a = a

    def log_prob(self, x):
        x2 = x if x.dim() == 2 else x.reshape(1, -1)
        w = x2 * x2 - 1.0
        return (-self.a) * (w * w).sum(dim=1)

    def gradient(self, x):
        # d(-log p)/dx, the sign binf's pdfs use (binf/pdf/__init__.py:190-191)
        return (4.0 * self.a) * x * (x * x - 1.0)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--chains', type=int, default=2048)
    ap.add_argument('--dims', type=int, default=64)
    ap.add_argument('--draws', type=int, default=400)
    ap.add_argument('--nsteps', type=int, default=10)
    ap.add_argument('--timestep', type=float, default=0.08)
    ap.add_argument('--a', type=float, default=2.0)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--graph', action='store_true',
                    help='capture the transition once as a HIP graph and replay it (same bits)')
    args = ap.parse_args(argv)

    dev = torch.device('cuda', torch.cuda.current_device())
    rng = DeviceRNG(args.seed, dev)
    state = rng.normal((args.chains, args.dims), dev)
    sampler = HMCSampler(DoubleWell(args.a), state, args.timestep, args.nsteps,
                         variable_name='x', rng=rng, graph=args.graph)
    right = torch.zeros((), dtype=torch.float64, device=dev)
    absx = torch.zeros((), dtype=torch.float64, device=dev)
    kept = 0
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.draws):
        x = sampler.sample()
        if i >= args.draws // 2:
            right += (x > 0).double().mean()
            absx += x.abs().mean()
            kept += 1
    torch.cuda.synchronize()
    print('{:.3f} ms per sample() ({})'.format((time.perf_counter() - t0) / args.draws * 1e3,
                                             'HIP graph replay' if sampler._graphs else 'eager launches'))
    print('acceptance rate          : {:.3f}'.format(float(sampler.acceptance_rate.mean())))
    print('fraction in the right well: {:.3f} (target 0.5)'.format(float(right) / kept))
    print('mean |x|                 : {:.3f} (wells at 1)'.format(float(absx) / kept))
    return sampler


if __name__ == '__main__':
    main()

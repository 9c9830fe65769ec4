#!/usr/bin/env python3
"""n Gibbs sweeps per launch (GibbsSampler.sample_n, csrc/gibbs_poly.hip) against
the loop of single sweeps, on the reference's example shape (K = 4 coefficients,
20 data points, HMC with 50 leapfrog steps or the reference's RWMC wiring).
Prints one JSON line per configuration; --out appends them to a file."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.example.misc import make_posterior
from binf_amd.example.samplers import make_hmc_sampler, make_sampler
from binf_amd.samplers import BinfState
from binf_amd.samplers.rng import DeviceRNG


def build(C, move, dev, seed=0, nsteps=50):
    np.random.seed(0)
    xs = np.linspace(-2, 2, 20)
    poly = np.polynomial.polynomial.polyval
    ys = np.random.normal(loc=poly(xs, np.array([2.0, -4.0, 1.0, 1.5])), scale=1.0 / np.sqrt(2.5))
    st = BinfState(dict(coefficients=torch.ones((C, 4), dtype=torch.float64, device=dev),
                        precision=torch.ones(C, dtype=torch.float64, device=dev)))
    post = make_posterior(xs, ys, poly)
    rng = DeviceRNG(seed, dev)
    if move == 'hmc':
        return make_hmc_sampler(post, 0.02, nsteps, st, rng=rng)
    return make_sampler(post, 0.1, st, rng=rng)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--chains', type=int, nargs='+', default=[4096, 65536])
    ap.add_argument('--n', type=int, default=500)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    lines = []
    for move in ('hmc', 'rwmc'):
        for C in a.chains:
            g1, g2, g3 = build(C, move, dev), build(C, move, dev), build(C, move, dev)
            g1.fused_sweep = False               # subsampler by subsampler (~8 launches per sweep)
            # same sweeps either way
            for _ in range(8):
                g1.sample()
                g3.sample()                      # the sweep as one launch
            g2.sample_n(8, record=False)
            same = torch.equal(g1.state.variables['coefficients'], g3.state.variables['coefficients'])
            same = same and torch.equal(g1.state.variables['coefficients'], g2.state.variables['coefficients']) \
                and torch.equal(g1.state.variables['precision'], g2.state.variables['precision'])
            t_loop = timed(lambda: [g1.sample() for _ in range(100)], 3) / 100
            t_one = timed(lambda: [g3.sample() for _ in range(100)], 3) / 100
            t_n = timed(lambda: g2.sample_n(a.n, thin=20), 3) / a.n
            L = 50 if move == 'hmc' else 0
            line = dict(move=move, chains=C, sweeps_per_launch=a.n, identical_states=bool(same),
                        loop_us_per_sweep=t_loop * 1e6, one_launch_per_sweep_us=t_one * 1e6,
                        launch_us_per_sweep=t_n * 1e6,
                        speedup=t_loop / t_n,
                        chain_leapfrog_steps_per_s=(C * L / t_n) if L else None,
                        chain_sweeps_per_s=C / t_n)
            print(json.dumps(line), flush=True)
            lines.append(line)
    if a.out:
        with open(a.out, 'a') as f:
            for l in lines:
                f.write(json.dumps(l) + '\n')


if __name__ == "__main__":
    main()

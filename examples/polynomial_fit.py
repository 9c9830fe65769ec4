#!/usr/bin/env python3
"""
The reference's example_script.py (polynomial fit: Gibbs over the polynomial
coefficients and the noise precision), run on an MI355X with many chains at
once.  Differences to the reference script, all deliberate:

* `--chains` independent chains instead of one (every state value is a
  [n_chains x ...] device tensor);
* the coefficients are sampled with HMC (50 leapfrog steps) instead of the
  random-walk sampler (pass --rwmc for the reference's own wiring);
* draws come from the device generator unless --host-rng is given (every rank
  uses the SAME seed and its shard's chain offset, so the result does not depend
  on how many GPUs the chains are spread over);
* samples are recorded in an on-device, thinned SampleStore (burn-in and
  thinning as in example_script.py:41) and gathered once at the end; with
  torch.distributed initialised the chains are sharded over the ranks;
* the sweeps run `--per-launch` at a time in one kernel launch
  (GibbsSampler.sample_n, bit-identical to that many gips.sample() calls;
  --per-launch 1 is the reference's loop, one sample() per iteration);
* no plotting (out of scope); a posterior summary is printed instead.

  python examples/polynomial_fit.py --chains 4096 --iterations 3000
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \\
      --master-addr 127.0.0.1 examples/polynomial_fit.py --chains 32768
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.dist import SampleStore, shard_chains, world
from binf_amd.example.misc import make_posterior
from binf_amd.example.samplers import make_hmc_sampler, make_sampler
from binf_amd.samplers import BinfState
from binf_amd.samplers.rng import DeviceRNG


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--chains', type=int, default=1024, help='total over all ranks')
    ap.add_argument('--iterations', type=int, default=3000)
    ap.add_argument('--burn-in', type=int, default=2000)
    ap.add_argument('--thin', type=int, default=20)
    ap.add_argument('--timestep', type=float, default=0.02)
    ap.add_argument('--nsteps', type=int, default=50)
    ap.add_argument('--rwmc', action='store_true', help="the reference's RWMC + Gamma wiring")
    ap.add_argument('--host-rng', action='store_true', help='np.random draws (parity mode, slow)')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--per-launch', type=int, default=500,
                    help='sweeps per kernel launch (1: one gips.sample() per iteration)')
    args = ap.parse_args(argv)

    if 'RANK' in os.environ and int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl')
    rank, ws = world()
    dev = torch.device('cuda', torch.cuda.current_device())
    start_chain, C = shard_chains(args.chains, rank, ws)

    # example_script.py:17-26
    np.random.seed(args.seed)
    n_data_points = 20
    real_coeffs = np.array([2.0, -4.0, 1.0, 1.5])
    real_precision = 2.5
    polynomial = np.polynomial.polynomial.polyval
    xses = np.linspace(-2, 2, n_data_points)
    ys = np.random.normal(loc=polynomial(xses, real_coeffs),
                          scale=1.0 / np.sqrt(real_precision))

    start = BinfState(dict(
        coefficients=torch.ones((C, 4), dtype=torch.float64, device=dev),
        precision=torch.ones(C, dtype=torch.float64, device=dev)))
    posterior = make_posterior(xses, ys, polynomial)
    # one seed for the whole run; the streams are keyed by the GLOBAL chain index
    rng = None if args.host_rng else DeviceRNG(args.seed, dev, chain_offset=start_chain)
    if args.rwmc:
        gips = make_sampler(posterior, 0.1, start, rng=rng)
    else:
        gips = make_hmc_sampler(posterior, args.timestep, args.nsteps, start,
                                **({} if rng is None else {'rng': rng}))

    n_keep = max(1, (args.iterations - args.burn_in + args.thin - 1) // args.thin)
    store_c = SampleStore(n_keep, C, 4, thin=args.thin, burn_in=args.burn_in, device=dev)
    store_p = SampleStore(n_keep, C, 1, thin=args.thin, burn_in=args.burn_in, device=dev)
    def report(i):
        if rank == 0:
            print('#### Gibbs sampling step {} ####'.format(i))
            stats = gips.last_draw_stats['coefficients']
            acc = stats.acceptance_rate if args.rwmc else stats.accepted.double()
            print('coefficient sampler acceptance: {:.3f}'.format(float(acc.mean())))

    if args.per_launch <= 1:
        for i in range(args.iterations):
            state = gips.sample()
            store_c.record(state.variables['coefficients'])
            store_p.record(state.variables['precision'])
            if i % 500 == 0 and i > 0:
                report(i)
    else:
        # the same sweeps, many per launch: burn-in unrecorded, then the draw the
        # store keeps first (sweep burn_in + 1), then blocks of whole thinning periods
        done = 0
        while done < min(args.burn_in, args.iterations):
            m = min(args.per_launch, min(args.burn_in, args.iterations) - done)
            gips.sample_n(m, record=False)
            done += m
            store_c.n_seen = store_p.n_seen = done
            report(done)
        if done < args.iterations:
            state = gips.sample()
            store_c.record(state.variables['coefficients'])
            store_p.record(state.variables['precision'])
            done += 1
        block = max(args.thin, args.per_launch // args.thin * args.thin)
        while done < args.iterations:
            m = min(block, args.iterations - done)
            rec = gips.sample_n(m, thin=args.thin)
            if rec is not None:
                store_c.extend(rec['coefficients'], n_sweeps=m)
                store_p.extend(rec['precision'].unsqueeze(-1), n_sweeps=m)
            done += m
            report(done)

    coeffs = store_c.gather(args.chains)           # [n_kept, chains, 4] on every rank
    prec = store_p.gather(args.chains)
    if rank == 0 and coeffs.shape[0] == 0:
        print('no draw kept: --iterations {} does not exceed --burn-in {}'.format(args.iterations, args.burn_in))
    elif rank == 0:
        c = coeffs.reshape(-1, 4)
        print('kept {} draws x {} chains'.format(coeffs.shape[0], coeffs.shape[1]))
        print('true coefficients     :', real_coeffs, ' precision', real_precision)
        print('posterior mean (coeff):', c.mean(0).cpu().numpy().round(3))
        print('posterior std  (coeff):', c.std(0).cpu().numpy().round(3))
        print('posterior mean (prec) : {:.3f}'.format(float(prec.mean())))
        # example_script.py:42 and :50-56 without the drawing: the log-probability of every kept
        # sample (one batched evaluation), the MAP estimate, and the posterior-predictive tube
        # over the data range -- predict() for 100 x 150 points x all kept samples in ONE launch
        # where the reference loops in Python (binf/example/plots.py:10-11)
        import time
        from binf_amd.example.misc import prediction_tube
        p = prec.reshape(-1)
        log_probs = posterior.log_prob(coefficients=c, precision=p)
        best = int(torch.argmax(log_probs))
        map_coeffs = c[best].cpu().numpy()
        print('MAP coefficients      :', map_coeffs.round(3), ' precision {:.3f}'.format(float(p[best])))
        space = np.linspace(-2, 2, 100)
        map_fit = polynomial(space, map_coeffs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tube = prediction_tube((c, p), polynomial, space, map_fit - 10, map_fit + 10, 150)
        dt = time.perf_counter() - t0
        print('prediction tube       : {} x {} points x {} samples in {:.1f} ms; 90 % interval at x = 0: '
              '[{:.2f}, {:.2f}], prediction {:.2f} (true curve {:.2f})'.format(
                  len(space), 150, c.shape[0], dt * 1e3, tube.lower[50], tube.upper[50],
                  tube.prediction[50], polynomial(space[50], real_coeffs)))
    return coeffs, prec


if __name__ == '__main__':
    main()

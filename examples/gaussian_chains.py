#!/usr/bin/env python3
"""
Many independent HMC chains on an isotropic Gaussian -- BASELINE config C2's
posterior (the reference's ``TestHO``, ``binf/pdf/__init__.py:181-191``) through
the reference's sampler surface::

    sampler = HMCSampler(pdf, state, timestep, nsteps, variable_name='x')
    for i in range(n): sampler.sample()              # binf style, one launch each
    draws = sampler.sample_n(n, thin=...)            # the same loop in one launch

With torch.distributed initialised the chains are sharded over the ranks and
the kept draws gathered once at the end (RCCL).

  python examples/gaussian_chains.py --chains 4096 --dims 1024 --draws 640
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \\
      --master-addr 127.0.0.1 examples/gaussian_chains.py --chains 32768
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.dist import gather_chains, shard_chains, world
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--chains', type=int, default=4096, help='total over all ranks')
    ap.add_argument('--dims', type=int, default=1024)
    ap.add_argument('--draws', type=int, default=640, help='transitions per chain')
    ap.add_argument('--thin', type=int, default=64)
    ap.add_argument('--nsteps', type=int, default=20)
    ap.add_argument('--timestep', type=float, default=0.05)
    ap.add_argument('--k', type=float, default=1.0)
    ap.add_argument('--x0', type=float, default=0.0)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args(argv)

    if 'RANK' in os.environ and int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl')
    rank, ws = world()
    dev = torch.device('cuda', torch.cuda.current_device())
    _, C = shard_chains(args.chains, rank, ws)

    rng = DeviceRNG(args.seed + 1000 * rank, dev)
    state = args.x0 + 3.0 * rng.normal((C, args.dims), dev)       # over-dispersed start
    sampler = HMCSampler(IsotropicGaussian(args.k, args.x0), state, args.timestep,
                         args.nsteps, variable_name='x', rng=rng)
    kept = []
    per_launch = max(args.thin, (64 // args.thin) * args.thin)
    done = 0
    while done < args.draws:
        n = min(per_launch, args.draws - done)
        out = sampler.sample_n(n, thin=args.thin)                 # [n // thin, C, D] or None
        if out is not None:
            kept.append(out)
        done += n
    draws = torch.cat(kept) if kept else torch.empty((0, C, args.dims), dtype=torch.float64, device=dev)
    # chains to dim 0 for the collective
    allc = gather_chains(draws.transpose(0, 1).contiguous(), args.chains).transpose(0, 1)
    if rank == 0:
        tail = allc[allc.shape[0] // 2:]
        print('kept {} draws x {} chains x {} dims'.format(*allc.shape))
        print('acceptance rate       : {:.3f}'.format(float(sampler.acceptance_rate.mean())))
        print('mean (target {:+.3f})  : {:+.4f}'.format(args.x0, float(tail.mean())))
        print('variance (target {:.3f}): {:.4f}'.format(1.0 / args.k, float(tail.var())))
    return allc


if __name__ == '__main__':
    main()

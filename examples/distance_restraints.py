#!/usr/bin/env python3
"""
BASELINE config C5: a chromatin-like bead chain (3 x n_beads coordinates)
restrained by noisy pairwise distances, many HMC chains at once.  The model is
written in the reference's plug-in shape -- a ForwardModel (coordinates -> all
n(n-1)/2 pair distances), a Gaussian ErrorModel on them, a Likelihood, a
Posterior with an isotropic Gaussian prior -- and sampled with HMCSampler; the
all-pairs force and the whole leapfrog integration run in fused HIP kernels
(the [3n x n(n-1)/2] Jacobian of the generic chain rule is never formed).
The reference has no code for this model (README.rst:9 only mentions the
application), so it is build-defined.

  python examples/distance_restraints.py --chains 256 --beads 256 --iterations 200
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.dist import SampleStore, shard_chains, world
from binf_amd.example.distance import make_distance_likelihood
from binf_amd.pdf import IsotropicGaussian
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--chains', type=int, default=256, help='total over all ranks')
    ap.add_argument('--beads', type=int, default=256)
    ap.add_argument('--iterations', type=int, default=200)
    ap.add_argument('--thin', type=int, default=10)
    ap.add_argument('--nsteps', type=int, default=20)
    ap.add_argument('--timestep', type=float, default=0.002)
    ap.add_argument('--precision', type=float, default=4.0)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--gibbs', action='store_true',
                    help='sample the noise precision too (Gibbs: HMC on the coordinates, conjugate '
                         'Gamma draw of one precision per chain)')
    args = ap.parse_args(argv)

    if 'RANK' in os.environ and int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl')
    rank, ws = world()
    dev = torch.device('cuda', torch.cuda.current_device())
    _, C = shard_chains(args.chains, rank, ws)
    n = args.beads

    # synthetic structure and noisy target distances (same on every rank)
    rs = np.random.RandomState(args.seed)
    truth = np.cumsum(rs.standard_normal((n, 3)), axis=0) * 0.5
    I, J = np.triu_indices(n, 1)
    d_true = np.sqrt(((truth[I] - truth[J]) ** 2).sum(1))
    ys = np.abs(d_true + rs.standard_normal(d_true.shape) / np.sqrt(args.precision))

    lik = make_distance_likelihood(ys, n)
    prior = IsotropicGaussian(0.01, 0.0, name='coordinates_prior', variable_name='coordinates')
    posterior = Posterior({lik.name: lik}, {prior.name: prior})
    cond = posterior.conditional_factory(precision=args.precision)

    rng = DeviceRNG(args.seed + 1 + 1000 * rank, dev)
    start = torch.from_numpy(truth.reshape(1, -1)).to(dev) + 0.3 * rng.normal((C, 3 * n), dev)
    n_keep = max(1, (args.iterations + args.thin - 1) // args.thin)
    store = SampleStore(n_keep, C, 3 * n, thin=args.thin, device=dev)
    if args.gibbs:
        # the reference's Gibbs scheme around the same kernels: coordinates | precision by HMC,
        # precision | coordinates by the conjugate Gamma draw (binf/example/samplers.py:27-51)
        from binf_amd.example.distance import make_restraint_gibbs_sampler
        from binf_amd.example.priors import GammaPrior
        from binf_amd.samplers import BinfState
        full = Posterior({lik.name: lik}, {prior.name: prior, 'precision_prior': GammaPrior(1.0, 0.2)})
        state = BinfState({'coordinates': start,
                           'precision': torch.full((C,), 1.0, dtype=torch.float64, device=dev)})
        gips = make_restraint_gibbs_sampler(full, args.timestep, args.nsteps, state, rng=rng,
                                            timestep_adaption_limit=args.iterations // 2)
        for i in range(args.iterations):
            st = gips.sample()
            store.record(st.variables['coordinates'])
        sampler = gips.subsamplers['coordinates']
        tau = st.variables['precision']
        if rank == 0:
            print('sampled precision          : {:.2f} +- {:.2f} over the chains (data generated '
                  'with {:.2f})'.format(float(tau.mean()), float(tau.std()), args.precision))
    else:
        sampler = HMCSampler(cond, start, args.timestep, args.nsteps,
                             variable_name='coordinates', rng=rng)
        for i in range(args.iterations):
            store.record(sampler.sample())
    kept = store.gather(args.chains)                                 # [n_kept, chains, 3n]
    if rank == 0:
        x = kept[-1].reshape(-1, n, 3)
        Id, Jd = torch.from_numpy(I).to(dev), torch.from_numpy(J).to(dev)
        d = (x[:, Id] - x[:, Jd]).pow(2).sum(-1).sqrt()
        rmsd = (d - torch.from_numpy(d_true).to(dev)).pow(2).mean().sqrt()
        print('kept {} draws x {} chains'.format(kept.shape[0], kept.shape[1]))
        print('acceptance rate            : {:.3f}'.format(float(sampler.acceptance_rate.mean())))
        print('distance RMSD to the truth : {:.3f} (noise sd {:.3f})'.format(
            float(rmsd), 1.0 / np.sqrt(args.precision)))
    return kept


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""
Headline benchmark: chain*leapfrog-steps per second on BASELINE config C2
(1024-d isotropic Gaussian, 4096 chains per GPU, 20 leapfrog steps, fp64).

A "step" is one HMC transition (one HMCSampler.sample() worth of work) over the
whole chain batch; by default 64 of them are issued per launch of the fused HIP
trajectory kernel (HMCSampler.sample_n, --fuse), every state still recorded,
through the C ABI.  Inputs (state, a pool of
pre-generated momentum / uniform draws) are resident in HBM before the timed
region.  Multi-GPU: chains are sharded (weak scaling, 4096 chains per GPU), no
collective in the data path; the sample gather is timed separately.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
      --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X spec, MI355X_MICROARCH.md "HBM3E peak BW"
HBM_COPY_GBS = 6290.0        # measured float4 copy ceiling, same table


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1024,
                    help='timed transitions (HMCSampler.sample() calls worth of work)')
    ap.add_argument('--warmup', type=int, default=128)
    ap.add_argument('--chains', type=int, default=4096, help='chains per GPU')
    ap.add_argument('--dims', type=int, default=1024)
    ap.add_argument('--nsteps', type=int, default=20, help='leapfrog steps')
    ap.add_argument('--timestep', type=float, default=0.05)
    ap.add_argument('--mode', default='exact', choices=['exact', 'fma'])
    ap.add_argument('--pool', type=int, default=16,
                    help='momentum-draw buffers cycled through (pool*C*D*8 B; '
                         '16 -> 512 MiB, larger than the 256 MiB Infinity Cache)')
    ap.add_argument('--fuse', type=int, default=64,
                    help='transitions per launch: 1 = one HMCSampler.sample() '
                         'per launch; n > 1 = HMCSampler.sample_n(n), the '
                         'persistent kernel (same draws, bit-identical results, '
                         'every transition\'s state still written to HBM)')
    ap.add_argument('--thin', type=int, default=1,
                    help='with --fuse > 1: record every thin-th state')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-mode', action='store_true',
                    help='skip the short extra measurement in the other arithmetic mode')
    ap.add_argument('--cpu-chains', type=int, default=64)
    ap.add_argument('--cpu-calls', type=int, default=2000,
                    help='sample() rounds of the CPU baseline (~10 s on the GPU box)')
    return ap.parse_args()


def cpu_baseline(D, L, dt, chains, calls):
    """The reference semantics (one chain per sampler, numpy fp64,
    hmc.py:136-164) via the numpy restatement, on ONE host core; plus the C
    restatement on all host cores as an extra figure."""
    from oracle import c_oracle
    from oracle import ref_numpy as R
    rs = np.random.RandomState(1234)
    q0 = rs.standard_normal((chains, D))
    samplers = [R.RefHMCSampler(R.GaussianPDF(1.0, 0.0), q0[c].copy(), dt, L,
                                variable_name='x') for c in range(chains)]
    np.random.seed(1000)
    for s in samplers[:4]:
        s.sample()
    t0 = time.perf_counter()
    for _ in range(calls):
        for s in samplers:
            s.sample()
    t = time.perf_counter() - t0
    out = {'value': chains * calls * L / t, 'unit': 'chain*leapfrog-steps/s',
           'cores': 1, 'kind': 'port',
           'sample': '%d chains x %d sample() calls, D=%d, L=%d, numpy '
                     'restatement of hmc.py:136-164, %.1f s'
                     % (chains, calls, D, L, t)}
    ncores = os.cpu_count() or 1
    Cc = 64 * ncores
    q = rs.standard_normal((Cc, D))
    p = rs.standard_normal((Cc, D))
    u = rs.uniform(size=Cc)
    c_oracle.hmc_sample_gauss(q, p, u, dt, L, nthreads=ncores)
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        c_oracle.hmc_sample_gauss(q, p, u, dt, L, nthreads=ncores)
    t = time.perf_counter() - t0
    out['c_port_all_cores'] = {'value': Cc * reps * L / t, 'cores': ncores,
                               'sample': '%d chains x %d calls, C restatement, '
                                         'OpenMP' % (Cc, reps)}
    return out


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed'
                     '.run --nproc-per-node %d' % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit('bench.py needs a GPU (binf_amd has no CPU path)')
    # a checkout without the built library: compile it (one rank per node), never fall back
    lib_path = os.path.join(ROOT, 'binf_amd', 'csrc', 'libbinf_hip.so')
    if not os.path.exists(lib_path):
        if local_rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        else:
            t_wait = time.time()
            size = -1
            while time.time() - t_wait < 900:   # present and no longer growing
                time.sleep(3)
                now = os.path.getsize(lib_path) if os.path.exists(lib_path) else -1
                if now > 0 and now == size:
                    break
                size = now
    # BINF_BENCH_BACKEND=gloo: rehearsal of the multi-rank control flow on a box
    # with fewer GPUs than ranks (ranks then share devices); real runs use
    # nccl (= RCCL on ROCm), one rank per GPU.
    backend = os.environ.get('BINF_BENCH_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == 'nccl' else local_rank % max(1, ndev)
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from binf_amd.pdf import IsotropicGaussian
    from binf_amd.samplers.hmc import HMCSampler

    C, D, L, dt = args.chains, args.dims, args.nsteps, args.timestep
    K, W, P = max(1, args.steps), max(0, args.warmup), max(1, args.pool)

    # synthetic inputs, resident in HBM before the timed region
    q0 = torch.from_numpy(
        np.random.RandomState(1234 + rank).standard_normal((C, D))).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + rank)
    p_pool = [torch.randn((C, D), dtype=torch.float64, device=dev, generator=gen)
              for _ in range(P)]
    u_pool = [torch.rand(C, dtype=torch.float64, device=dev, generator=gen)
              for _ in range(P)]

    sampler = HMCSampler(IsotropicGaussian(1.0, 0.0), q0, dt, L,
                         variable_name='x', mode=args.mode)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    F = max(1, args.fuse)
    if F > 1:
        # one step is still ONE transition; they are issued F per launch (a
        # step count that is not a multiple of F ends with one shorter launch)
        nchunk = max(2, P // F)
        p_chunks = [torch.randn((F, C, D), dtype=torch.float64, device=dev,
                                generator=gen) for _ in range(nchunk)]
        u_chunks = [torch.rand((F, C), dtype=torch.float64, device=dev,
                               generator=gen) for _ in range(nchunk)]
        del p_pool, u_pool

        def run(nsteps, sampler=sampler):
            i = 0
            while nsteps > 0:
                n = min(F, nsteps)
                sampler.sample_n(n, thin=min(args.thin, n), p0=p_chunks[i % nchunk][:n],
                                 u=u_chunks[i % nchunk][:n], record=True)
                nsteps -= n
                i += 1
    else:
        def run(nsteps, sampler=sampler):
            for i in range(nsteps):
                sampler.sample(p0=p_pool[i % P], u=u_pool[i % P])

    run(W)
    barrier()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(K)
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)              # same stream as the launches

    acc_rate = float(sampler.acceptance_rate.mean())
    gather_ms = None

    # Outside the metric: the same workload in the other arithmetic mode ('fma'
    # contracts each multiply-add; within 1e-10 of 'exact', not bit-identical).
    other = None
    if world == 1 and not args.no_other_mode:
        om = 'fma' if args.mode == 'exact' else 'exact'
        s2 = HMCSampler(IsotropicGaussian(1.0, 0.0), q0, dt, L, variable_name='x', mode=om)
        K2 = min(K, 4 * F)
        run(min(W, F), s2)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        run(K2, s2)
        e1.record()
        torch.cuda.synchronize()
        t2 = e0.elapsed_time(e1) * 1e-3 / K2
        other = {'mode': om, 'value': C * L / t2, 'avg_transition_us': t2 * 1e6,
                 'roofline_frac': (24.0 * D + 25.0) * C / t2 / 1e9 / HBM_PEAK_GBS,
                 'steps': K2, 'note': 'device time (HIP events), not the headline value'}
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64,
                            device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0])
        # the only collective of the path: gather one recorded draw (RCCL)
        from binf_amd.dist import gather_chains
        state = sampler.state if backend == 'nccl' else sampler.state.cpu()
        gather_chains(state)
        barrier()
        t1 = time.perf_counter()
        for _ in range(5):
            gather_chains(state)
        barrier()
        gather_ms = (time.perf_counter() - t1) / 5 * 1e3

    if rank == 0:
        steps_total = float(world) * C * L * K
        value = steps_total / elapsed
        bytes_per_transition = (24.0 * D + 25.0) * C
        n_launches = (K + F - 1) // F
        launch_s = dev_ms * 1e-3 / n_launches             # per kernel launch
        trans_s = dev_ms * 1e-3 / K                       # per transition
        bytes_per_launch = bytes_per_transition * min(F, K)   # a full launch
        # algorithmic bytes of the timed region / device time of the timed
        # region (= bytes per launch / average launch duration)
        achieved = bytes_per_transition * K / (dev_ms * 1e-3) / 1e9
        # HBM traffic from the committed PMC summary of this exact configuration
        traffic, traffic_src = None, None
        for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json')),
                           reverse=True):
            try:
                pm = json.load(open(path))
                c = pm['config']
                if (c['chains'], c['dims'], c['nsteps'], c['fuse'], c['thin'], c['mode']) == \
                        (C, D, L, F, args.thin if F > 1 else 1, args.mode):
                    traffic = pm['hbm_bytes_per_transition'] * min(F, K)
                    traffic_src = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, ' \
                                  'separate passes, FETCH x2)' % os.path.basename(path)
                    break
            except (OSError, ValueError, KeyError):
                continue
        res = {
            'metric': 'chain*leapfrog-steps/sec, 1024-d Gaussian',
            'value': value,
            'unit': 'chain*leapfrog-steps/s',
            'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': elapsed / K * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': 'C2: %d-d isotropic Gaussian (k=1, x0=0), '
                                   '%d chains/GPU, %d leapfrog steps, dt=%g, '
                                   'fused HMC transition, mode=%s, %d '
                                   'transition(s) per launch'
                                   % (D, C, L, dt, args.mode, F),
                       'chains_per_gpu': C, 'n_dims': D, 'leapfrog_steps': L,
                       'parallelism': 'chains sharded x%d, no data-path '
                                      'collective' % world,
                       'draw_pool_buffers': P},
            'acceptance_rate': acc_rate,
            'roofline': {'bound': 'hbm', 'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS,
                         'frac_of_measured_copy_ceiling': achieved / HBM_COPY_GBS,
                         'traffic': traffic, 'traffic_source': traffic_src,
                         'per': 'kernel launch = %d transition(s), each one '
                                'HMCSampler.sample() worth of work' % F,
                         'kernel': 'hmc_gauss_persist_kernel',
                         'transitions_per_launch': F,
                         'states_recorded': 'every transition' if F == 1
                         else 'every %d. transition' % args.thin,
                         'algorithmic_bytes_per_launch': bytes_per_launch,
                         'avg_launch_us': launch_s * 1e6,
                         'avg_transition_us': trans_s * 1e6},
        }
        if gather_ms is not None:
            res['sample_gather_ms'] = gather_ms
        if other is not None:
            res['other_mode'] = other
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(D, L, dt, args.cpu_chains,
                                               args.cpu_calls)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

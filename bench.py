#!/usr/bin/env python3
"""
Headline benchmark: chain*leapfrog-steps per second on BASELINE config C2
(1024-d isotropic Gaussian, 4096 chains per GPU, 20 leapfrog steps, fp64).

A "step" is ONE launch of the fused HIP trajectory kernel through the C ABI:
HMCSampler.sample_n(F) = F (default 64) consecutive HMC transitions of every
chain -- the reference's `for i in range(F): sampler.sample()` loop
(example_script.py:33-34 around binf/samplers/hmc.py:136-164) -- with the state
after EVERY transition written to a record buffer in HBM.  Inputs (start state,
pre-generated momentum / uniform draws) and the record buffers are resident in
HBM and allocated before the timed region; each timed launch reads draws that no
earlier launch in the run has touched recently (>= 2 GiB of other traffic in
between, the Infinity Cache is 256 MiB).

Multi-GPU: chains are sharded (weak scaling, 4096 chains per GPU; --scaling strong:
4096 in all), no collective in the data path; the RCCL gather of one recorded draw
is timed separately.  With N > 1 every rank also runs the sharded C4 (Gibbs-within-HMC,
polynomial) and C5 (pair-distance) legs -- scripts/bench_legs.py: per-rank shards by
DeviceRNG.for_shard, SampleStore recording, a timed gather(dst=0), per-rank self-check
fields -- and, in a weak run, the fixed-size C2 job sharded over the same ranks
(extra.C2_strong): one line holds every figure north_star names.

  python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no torch.distributed environment, bench.py starts the N
ranks itself (python -m torch.distributed.run ... bench.py, as child processes,
before this process touches the GPU) and relays rank 0's line.  It can equally be
started by torch.distributed.run directly.

Prints ONE JSON line on rank 0.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X spec, MI355X_MICROARCH.md "HBM3E peak BW"
HBM_COPY_GBS = 6290.0        # measured float4 copy ceiling, same table
# FP64 vector peak without FMA contraction: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz
# = 39.3e12 lane-operations/s (half of the 78.6 TFLOP/s FMA figure)
VALU_PEAK_LANEOPS = 39.3e12
PERSIST_KERNEL = 'hmc_gauss_persist_kernel'


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20,
                    help='timed steps; one step = one sample_n(--fuse) launch')
    ap.add_argument('--warmup', type=int, default=5, help='untimed steps')
    ap.add_argument('--settle-ms', type=float, default=300.0,
                    help='untimed pre-run of the same workload before the warm-up steps: the '
                         'chip needs ~50 ms of sustained load before its clock / power '
                         'controller settles (the first ~20 launches of a cold run take up to '
                         '1.5x the settled time; profiles/r02_settle_notes.md); 0 = none')
    ap.add_argument('--chains', type=int, default=4096,
                    help='chains per GPU (--scaling weak) / chains in all (--scaling strong)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='weak (primary, SURVEY 8(e)): --chains per GPU, the job grows with N; '
                         'strong (secondary): --chains in all, sharded over the N ranks '
                         '(4096 -> 512 per GPU at N = 8, a chain spread over several waves); '
                         'the C4 / C5 legs then shard their BASELINE totals (32768 / 2048 chains)')
    ap.add_argument('--no-legs', action='store_true',
                    help='with --gpus N > 1: skip the sharded C4 / C5 legs')
    ap.add_argument('--legs-timeout', type=float, default=300.0,
                    help='with --gpus N > 1: seconds the sharded legs may take before rank 0 prints '
                         'the headline line without them (LineGuard)')
    ap.add_argument('--dims', type=int, default=1024)
    ap.add_argument('--nsteps', type=int, default=20, help='leapfrog steps')
    ap.add_argument('--timestep', type=float, default=0.05)
    ap.add_argument('--mode', default='exact', choices=['exact', 'fma'])
    ap.add_argument('--fuse', type=int, default=64,
                    help='transitions per step: n > 1 = HMCSampler.sample_n(n), '
                         'one launch of the persistent kernel; 1 = one '
                         'HMCSampler.sample() per step (bit-identical results)')
    ap.add_argument('--thin', type=int, default=1,
                    help='with --fuse > 1: record every thin-th state')
    ap.add_argument('--draw-buffers', type=int, default=3,
                    help='draw buffers [fuse, C, D] cycled through (>= 2)')
    ap.add_argument('--survey-draws', type=int, default=1,
                    help='draw buffers generated on the host with the SURVEY.md 8(d) seeds '
                         '(the rest: torch.randn on the device)')
    ap.add_argument('--sustain-ms', type=float, default=1000.0,
                    help='after the timed steps: the same launches for this long, reported as '
                         'value_sustained (0 = skip)')
    ap.add_argument('--no-single-call', action='store_true',
                    help='skip the short one-transition-per-launch measurement (roofline.single_sample_call)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-mode', action='store_true',
                    help='skip the short extra measurement in the other arithmetic mode')
    ap.add_argument('--no-extra', action='store_true',
                    help='skip the C3 / C5 / device-RNG sub-results')
    ap.add_argument('--no-pmc', action='store_true',
                    help='do not measure HBM traffic with rocprofv3 --pmc child '
                         'runs; use the committed profile of this configuration')
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-chains', type=int, default=64)
    ap.add_argument('--cpu-calls', type=int, default=2000,
                    help='sample() rounds of the CPU baseline (~10 s on the GPU box)')
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------
# CPU baseline (the only place the oracle is used here: as the thing compared
# against, never inside the measured path)
# ---------------------------------------------------------------------------
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


# ---------------------------------------------------------------------------
# launching
# ---------------------------------------------------------------------------
def ensure_library():
    """A checkout without the built library: compile it (hipcc needs no GPU and
    this runs before any GPU call).  build() writes to a temporary name and
    renames, so a rank that sees the file sees a complete one."""
    lib_path = os.path.join(ROOT, 'binf_amd', 'csrc', 'libbinf_hip.so')
    if os.path.exists(lib_path):
        return
    if int(os.environ.get('LOCAL_RANK', '0')) == 0:
        import __graft_entry__
        __graft_entry__.build()
        return
    t_wait = time.time()
    while not os.path.exists(lib_path):
        if time.time() - t_wait > 900:
            sys.exit('bench.py: %s did not appear (local rank 0 builds it)' % lib_path)
        time.sleep(1)


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args, argv):
    """--gpus N > 1 without a torch.distributed environment: start the N ranks
    as CHILD processes (this process has not touched the GPU and never will)."""
    backend = os.environ.get('BINF_BENCH_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()       # does not initialise the GPU
    if backend == 'nccl' and ndev < args.gpus:
        sys.stderr.write('bench.py --gpus %d: only %d GPU(s) visible on this node '
                         '(one rank per GPU; BINF_BENCH_BACKEND=gloo rehearses the '
                         'control flow with ranks sharing devices)\n'
                         % (args.gpus, ndev))
        sys.exit(2)
    ensure_library()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '1')
    proc = subprocess.run(cmd, env=env)
    sys.exit(proc.returncode)


# ---------------------------------------------------------------------------
# HBM traffic from the PMC counters, measured in THIS run: two child runs of the
# same configuration under rocprofv3 (FETCH_SIZE and WRITE_SIZE in separate
# passes), started before this process initialises the GPU.
# ---------------------------------------------------------------------------
def under_profiler():
    if 'rocprof' in os.environ.get('LD_PRELOAD', ''):
        return True
    return any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ)


def pmc_traffic_live(args):
    """-> (hbm bytes per launch, description) or (None, reason)."""
    rocprof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(rocprof):
        return None, 'rocprofv3 not found'
    child = [sys.executable, os.path.abspath(__file__), '--pmc-child',
             '--chains', str(args.chains), '--dims', str(args.dims),
             '--nsteps', str(args.nsteps), '--timestep', repr(args.timestep),
             '--mode', args.mode, '--fuse', str(args.fuse), '--thin', str(args.thin),
             '--steps', '3', '--warmup', '1']
    kb = {}
    tmp = tempfile.mkdtemp(prefix='binf_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(tmp, counter)
            cmd = [rocprof, '--kernel-trace', '--pmc', counter, '--output-format', 'csv',
                   '-d', out, '--'] + child
            try:
                r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE,
                                   stderr=subprocess.PIPE, timeout=150)
            except subprocess.TimeoutExpired:
                return None, 'rocprofv3 --pmc %s child timed out' % counter
            if r.returncode != 0:
                return None, 'rocprofv3 --pmc %s child failed (rc %d)' % (counter, r.returncode)
            vals = []
            for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
                for row in csv.DictReader(open(f)):
                    if PERSIST_KERNEL in row.get('Kernel_Name', '') and \
                            row.get('Counter_Name') == counter:
                        vals.append(float(row['Counter_Value']))
            if not vals:
                return None, 'no %s rows for %s in the rocprofv3 output' % (counter, PERSIST_KERNEL)
            kb[counter] = (sum(vals) / len(vals), len(vals))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # gfx950: FETCH_SIZE tallies reads at half size (MI355X_MICROARCH.md, HBM /
    # rocprofv3 section; calibration in profiles/r01_h_pmc_calibration.json:
    # 0.5000 / 1.0000 on a kernel of known traffic); both counters are in KiB
    nbytes = (2.0 * kb['FETCH_SIZE'][0] + kb['WRITE_SIZE'][0]) * 1024.0
    src = ('measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE '
           '(separate child passes of this configuration, mean over %d / %d dispatches of %s; '
           'FETCH_SIZE x2 (gfx950 correction), KiB -> B)'
           % (kb['FETCH_SIZE'][1], kb['WRITE_SIZE'][1], PERSIST_KERNEL))
    return nbytes, src


def kernel_stats_live():
    """rocprofv3 --kernel-trace --stats over scripts/bench_extra.py --roofline-child
    (the C3 / C5 / C1 kernels the sub-results price) -> ({kernel: (calls, avg ns)}, text)."""
    rocprof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(rocprof):
        return None, 'rocprofv3 not found'
    tmp = tempfile.mkdtemp(prefix='binf_kt_', dir='/tmp')
    cmd = [rocprof, '--kernel-trace', '--stats', '--output-format', 'csv', '-d', tmp, '--',
           sys.executable, os.path.join(ROOT, 'scripts', 'bench_extra.py'), '--roofline-child']
    try:
        try:
            r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'),
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
        except subprocess.TimeoutExpired:
            return None, 'rocprofv3 --kernel-trace child timed out'
        if r.returncode != 0:
            return None, 'rocprofv3 --kernel-trace child failed (rc %d)' % r.returncode
        stats = {}
        for f in glob.glob(os.path.join(tmp, '**', '*kernel_stats.csv'), recursive=True):
            for row in csv.DictReader(open(f)):
                stats[row['Name']] = (int(row['Calls']), float(row['AverageNs']))
        return (stats, 'ok') if stats else (None, 'no kernel_stats.csv in the rocprofv3 output')
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_traffic_committed(C, D, L, F, thin, mode):
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json')),
                       reverse=True):
        try:
            pm = json.load(open(path))
            c = pm['config']
            if (c['chains'], c['dims'], c['nsteps'], c['fuse'], c['thin'], c['mode']) == \
                    (C, D, L, F, thin, mode):
                return pm['hbm_bytes_per_transition'] * F, \
                    'profiles/%s (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summary of ' \
                    'this configuration, separate passes, FETCH x2)' % os.path.basename(path)
        except (OSError, ValueError, KeyError):
            continue
    return None, None


# ---------------------------------------------------------------------------
# stdout carries ONE line.  Libraries write there too -- RCCL a version banner when its first
# communicator is made, gloo its connection messages -- so a rank hands its C-level stdout over to
# stderr as it starts and writes the JSON line to the descriptor it kept.
_LINE_FD = None


def claim_stdout():
    global _LINE_FD
    if _LINE_FD is None:
        sys.stdout.flush()
        _LINE_FD = os.dup(1)
        os.dup2(2, 1)


def emit(res):
    line = (json.dumps(res) + '\n').encode()
    if _LINE_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_LINE_FD, line)


# ---------------------------------------------------------------------------
class LineGuard:
    """The optional multi-rank sections (the sharded C4 / C5 legs) run under this guard: if
    they have not finished after ``seconds`` -- one rank failed on its own and the others
    wait in a collective for it -- rank 0 prints the headline line it already holds, with
    the reason in place of the legs, and every rank leaves with exit code 0.  A hang there
    costs the legs, never the N-GPU value."""

    def __init__(self, res, seconds, what='the sharded legs'):
        import threading
        self.res, self.seconds, self.what = res, seconds, what
        self.lock = threading.Lock()
        self.done = False
        self.timer = threading.Timer(seconds, self._fire)
        self.timer.daemon = True
        self.timer.start()

    def _fire(self):
        with self.lock:
            if self.done:
                return
            self.done = True
            if self.res is not None:
                self.res['extra'] = {'error': '%s did not finish within %g s (a rank failed or a '
                                              'collective hung); every field of the headline above '
                                              'was complete before they started'
                                              % (self.what, self.seconds)}
                emit(self.res)
            else:
                time.sleep(3.0)                     # rank 0 prints first
            sys.stdout.flush()
            os._exit(0)

    def finish(self):
        with self.lock:                             # blocks for good if the timer is printing
            self.done = True
        self.timer.cancel()


# ---------------------------------------------------------------------------
def dry_run(args, rank, world):
    """BINF_BENCH_DRYRUN=1: the multi-rank control flow without a GPU (CPU test
    suite): rendezvous, barrier, MAX over ranks, the sample gather -- no sampling."""
    import torch.distributed as dist
    from binf_amd.dist import gather_chains, shard_chains
    if world > 1:
        dist.init_process_group('gloo')
        dist.barrier()
    elapsed = 1.0 + rank
    gathered = None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0])
        total = args.chains * world if args.scaling == 'weak' else args.chains
        _, count = shard_chains(total, rank, world)
        state = torch.full((count, 4), float(rank), dtype=torch.float64)
        gathered = gather_chains(state, total)
        dist.barrier()
    # the sharded legs' control flow (scripts/bench_legs.py: barrier-bracketed sweeps,
    # SampleStore recording, MAX over ranks, gather to rank 0, per-rank self-check
    # fields) with a stand-in leg on host tensors
    from scripts import bench_legs
    comm = bench_legs.Comm(dist if world > 1 else None, 'gloo' if world > 1 else None,
                           torch.device('cpu'))
    start, count = shard_chains(args.chains if args.scaling == 'strong' else args.chains * world,
                                rank, world)
    res = None
    if rank == 0:
        res = {'dry_run': True, 'n_gpus': world, 'steps': args.steps,
               'warmup': args.warmup, 'max_elapsed': elapsed,
               'gathered_rows': None if gathered is None else int(gathered.shape[0]),
               'scaling': args.scaling, 'shard': [start, count], 'value': None}
    guard = LineGuard(res, args.legs_timeout) if world > 1 else None
    if os.environ.get('BINF_BENCH_DRYRUN_HANG_RANK') == str(rank):
        time.sleep(3600)                            # test hook: this rank never reaches the leg
    leg = bench_legs.StandInLeg(comm, chains_per_gpu=args.chains, scaling=args.scaling)
    legs = {'stand_in': bench_legs.run_leg(leg, comm, sweeps=6, warm=1, thin=2, settle_s=0.0)}
    if guard is not None:
        guard.finish()
    if rank == 0:
        res['extra'] = legs
        emit(res)
    if world > 1:
        dist.destroy_process_group()


def c2_strong_leg(args, comm, dev, F, thin, D, L, launches=10):
    """The C2 job at FIXED size -- args.chains chains in all -- sharded over the ranks of this run
    (512 per GPU at N = 8: a chain spread over several waves, csrc/hmc_gauss_split.hip): same
    launches as the headline (sample_n(F), every thin-th state recorded, draws resident in HBM),
    wall clock between two barriers, MAX over ranks."""
    from binf_amd.dist import shard_chains
    from binf_amd.pdf import IsotropicGaussian
    from binf_amd.samplers.hmc import HMCSampler
    total = args.chains
    off, Cs = shard_chains(total, comm.rank, comm.world)
    if Cs < 1:
        raise ValueError('%d chains leave rank %d without one' % (total, comm.rank))
    q0 = torch.from_numpy(np.ascontiguousarray(
        np.random.RandomState(1234).standard_normal((total, D))[off:off + Cs])).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7000 + comm.rank)
    p = [torch.randn((F, Cs, D), dtype=torch.float64, device=dev, generator=gen) for _ in range(2)]
    u = [torch.rand((F, Cs), dtype=torch.float64, device=dev, generator=gen) for _ in range(2)]
    rec = torch.empty((F // thin, Cs, D), dtype=torch.float64, device=dev)
    s = HMCSampler(IsotropicGaussian(1.0, 0.0), q0, args.timestep, L, variable_name='x', mode=args.mode)
    t_s = time.perf_counter()
    i = 0
    while time.perf_counter() - t_s < 0.15:            # settle
        s.sample_n(F, thin=thin, p0=p[i % 2], u=u[i % 2], out=rec)
        i += 1
        torch.cuda.synchronize()
    comm.barrier()
    t0 = time.perf_counter()
    for j in range(launches):
        s.sample_n(F, thin=thin, p0=p[j % 2], u=u[j % 2], out=rec)
    comm.barrier()
    mine = time.perf_counter() - t0
    elapsed = comm.max(mine)
    ranks = comm.all_gather_object({'rank': comm.rank, 'chain_offset': off, 'chains': Cs, 'elapsed_s': mine,
                                    'value': Cs * L * F * launches / mine,
                                    'acceptance': float(s.acceptance_rate.mean())})
    return {'workload': 'C2 strong scaling: %d chains IN ALL over %d GPUs (%d on rank 0), sample_n(%d) x %d '
                        'launches, every %s state recorded' % (total, comm.world, ranks[0]['chains'], F, launches,
                                                              'transition\'s' if thin == 1 else '%d.' % thin),
            'chain_leapfrog_steps_per_s': total * L * F * launches / elapsed,
            'chains_total': total, 'n_gpus': comm.world, 'scaling': 'strong',
            'us_per_transition': elapsed / (launches * F) * 1e6,
            'sum_of_rank_values': sum(r['value'] for r in ranks), 'ranks': ranks,
            'timing': 'wall clock between two barriers, MAX over ranks'}


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC only on this pool (RCCL)
    in_dist = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ
    if args.gpus > 1 and not in_dist:
        self_launch(args, argv)
    claim_stdout()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    args.gpus = world
    # BINF_BENCH_FORCE_DIST=1: take the N > 1 control flow (process group, per-rank fields,
    # gathers, the sharded C4 / C5 / strong-C2 legs) with however many ranks there are --
    # with ONE rank on a one-GPU box this drives every collective of the multi-GPU run
    # through RCCL itself (profiles/r04_n_*), which a gloo rehearsal cannot
    multi = world > 1 or os.environ.get('BINF_BENCH_FORCE_DIST') == '1'
    if multi and world == 1:
        os.environ['BINF_DIST_NO_SHORTCUT'] = '1'      # binf_amd/dist.py: no one-rank short cut
    if multi and not in_dist:
        import socket
        sock = socket.socket()
        sock.bind(('127.0.0.1', 0))
        os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
                          MASTER_PORT=str(sock.getsockname()[1]))
        sock.close()
    if os.environ.get('BINF_BENCH_DRYRUN') == '1':
        return dry_run(args, rank, world)
    ensure_library()

    from binf_amd.dist import shard_chains
    D, L, dt = args.dims, args.nsteps, args.timestep
    # weak: --chains per GPU; strong: --chains in all, this rank's contiguous block
    C_total = args.chains * world if args.scaling == 'weak' else args.chains
    chain_offset, C = shard_chains(C_total, rank, world)
    if C < 1:
        sys.exit('bench.py: --scaling strong with %d chains leaves rank %d without a chain'
                 % (C_total, rank))
    K, W = max(1, args.steps), max(0, args.warmup)
    F = max(1, args.fuse)
    thin = min(max(1, args.thin), F)
    NB = max(2, args.draw_buffers)

    # HBM traffic of the dominant kernel by PMC, measured live in child runs --
    # BEFORE this process initialises the GPU (rank 0 of a single-GPU run only)
    live_traffic, live_src = None, 'not attempted'
    if not multi and F > 1 and not args.pmc_child and not args.no_pmc:
        if under_profiler():
            live_src = 'this process itself runs under a profiler'
        else:
            t_p = time.perf_counter()
            live_traffic, live_src = pmc_traffic_live(args)
            if live_traffic is not None:
                live_src += '; %.0f s' % (time.perf_counter() - t_p)
    kstats, kstats_src = None, 'not attempted'
    if not multi and not args.pmc_child and not args.no_pmc and not args.no_extra:
        if under_profiler():
            kstats_src = 'this process itself runs under a profiler'
        else:
            kstats, kstats_src = kernel_stats_live()

    if not torch.cuda.is_available():
        sys.exit('bench.py needs a GPU (binf_amd has no CPU path)')
    # BINF_BENCH_BACKEND=gloo: rehearsal of the multi-rank control flow on a box
    # with fewer GPUs than ranks (ranks then share devices); real runs use
    # nccl (= RCCL on ROCm), one rank per GPU.
    backend = os.environ.get('BINF_BENCH_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()
    if backend == 'nccl' and world > ndev:
        sys.exit('bench.py: %d ranks but only %d GPU(s) visible (one rank per GPU)'
                 % (world, ndev))
    dev_index = local_rank if backend == 'nccl' else local_rank % max(1, ndev)
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    dist = None
    if multi:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from binf_amd.pdf import IsotropicGaussian
    from binf_amd.samplers.hmc import HMCSampler

    # synthetic inputs, resident in HBM before the timed region.  q0 and the FIRST draw
    # buffer follow SURVEY.md 8(d) literally -- q0 = RandomState(1234), call i draws
    # p0 = RandomState(1000 + i).standard_normal((C, D)), u = RandomState(2000 + i).uniform(C)
    # (rank r adds 100000 r to every seed) -- generated on the host before the timed
    # region; the other buffers are filled on the device (host generation of a buffer
    # takes ~6 s).  --survey-draws 0: all buffers on the device.
    if args.scaling == 'weak':
        q0 = torch.from_numpy(
            np.random.RandomState(1234 + 100000 * rank).standard_normal((C, D))).to(dev)
    else:       # one job whatever N: the rows of the whole start state this rank owns
        q0 = torch.from_numpy(np.ascontiguousarray(
            np.random.RandomState(1234).standard_normal((C_total, D))[chain_offset:chain_offset + C]
        )).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + rank)
    p_bufs, u_bufs = [], []
    n_survey = min(NB, max(0, args.survey_draws)) if not args.pmc_child else 0
    for b in range(NB):
        if b < n_survey:
            pb = torch.empty((F, C, D), dtype=torch.float64, device=dev)
            ub = torch.empty((F, C), dtype=torch.float64, device=dev)
            for i in range(F):
                call = b * F + i + 100000 * rank
                pb[i].copy_(torch.from_numpy(np.random.RandomState(1000 + call).standard_normal((C, D))))
                ub[i].copy_(torch.from_numpy(np.random.RandomState(2000 + call).uniform(size=C)))
        else:
            pb = torch.randn((F, C, D), dtype=torch.float64, device=dev, generator=gen)
            ub = torch.rand((F, C), dtype=torch.float64, device=dev, generator=gen)
        p_bufs.append(pb)
        u_bufs.append(ub)
    nrec = F // thin
    # record buffers (every thin-th state of a step), preallocated: two, used in turn
    rec_bufs = [torch.empty((nrec, C, D), dtype=torch.float64, device=dev)
                for _ in range(2)] if F > 1 else None

    def make_sampler(mode):
        return HMCSampler(IsotropicGaussian(1.0, 0.0), q0, dt, L,
                          variable_name='x', mode=mode)

    def run(sampler, first, nsteps):
        """steps first .. first+nsteps-1 of the run (the draw buffer of a step
        is first-use or last used >= NB-1 steps = >= 2 GiB of traffic ago)"""
        for i in range(first, first + nsteps):
            b = i % NB
            if F > 1:
                sampler.sample_n(F, thin=thin, p0=p_bufs[b], u=u_bufs[b],
                                 record=True, out=rec_bufs[i % 2])
            else:
                sampler.sample(p0=p_bufs[b][0], u=u_bufs[b][0])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.pmc_child:
        s = make_sampler(args.mode)
        run(s, 0, W + K)
        torch.cuda.synchronize()
        return

    sampler = make_sampler(args.mode)
    # settle: the same launches, untimed, until the clock / power controller has
    # converged (not part of the W warm-up steps; reported in the JSON line)
    n_settle = 0
    cold_us = None
    if args.settle_ms > 0:
        # the first launches of the run, one event each: the ramp the settle phase
        # exists for (reported, so that the cold-start reading is in the same line)
        cev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        cev[0].record()
        for i in range(20):
            run(sampler, i, 1)
            cev[i + 1].record()
        torch.cuda.synchronize()
        cold_us = [round(cev[i].elapsed_time(cev[i + 1]) * 1e3, 1) for i in range(20)]
        n_settle = 20
        t_s = time.perf_counter()
        while (time.perf_counter() - t_s) * 1e3 < args.settle_ms:
            run(sampler, n_settle, 4)
            torch.cuda.synchronize()
            n_settle += 4
    run(sampler, n_settle, W)
    barrier()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)] if K <= 512 else None
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                                # same stream as the launches
    if evs is None:
        run(sampler, n_settle + W, K)
    else:                                       # one event per launch: the spread
        evs[0].record()
        for i in range(K):
            run(sampler, n_settle + W + i, 1)
            evs[i + 1].record()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    my_elapsed = elapsed
    dev_ms = ev0.elapsed_time(ev1)
    per_launch_us = None
    if evs is not None:
        per_launch_us = sorted(evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(K))

    # the same launches for >= --sustain-ms more (device time, HIP events): the figure a long
    # run sees, next to the K timed steps
    sustained = None
    if args.sustain_ms > 0:
        s0 = torch.cuda.Event(enable_timing=True)
        s1 = torch.cuda.Event(enable_timing=True)
        n_sus, first = 0, n_settle + W + K
        t_s = time.perf_counter()
        s0.record()
        while (time.perf_counter() - t_s) * 1e3 < args.sustain_ms:
            run(sampler, first + n_sus, 8)
            n_sus += 8
            torch.cuda.synchronize()
        s1.record()
        torch.cuda.synchronize()
        sus_ms = s0.elapsed_time(s1)
        sustained = {'launches': n_sus, 'ms': sus_ms,
                     'value_per_gpu': C * L * F * n_sus / (sus_ms * 1e-3)}

    # the single sample() call GibbsSampler.sample() drives (gibbs.py:148): one transition per
    # launch, same kernel family -- reported beside the F-transition launch
    single = None
    if F > 1 and not multi and not args.pmc_child and not args.no_single_call:
        s1 = make_sampler(args.mode)
        for i in range(min(8, F)):
            s1.sample(p0=p_bufs[0][i], u=u_bufs[0][i])
        torch.cuda.synchronize()
        g0 = torch.cuda.Event(enable_timing=True)
        g1 = torch.cuda.Event(enable_timing=True)
        n1 = min(F, 48)
        g0.record()
        for i in range(n1):
            s1.sample(p0=p_bufs[1][i], u=u_bufs[1][i])
        g1.record()
        torch.cuda.synchronize()
        t1 = g0.elapsed_time(g1) * 1e-3 / n1
        single = {'us_per_call': t1 * 1e6, 'chain_leapfrog_steps_per_s': C * L / t1,
                  'frac': (24.0 * D + 25.0) * C / t1 / 1e9 / HBM_PEAK_GBS,
                  'what': 'HMCSampler.sample(): ONE transition per launch (the call '
                          'GibbsSampler.sample() makes), device time over %d calls; same contract '
                          'bytes per call as one transition of the headline launch' % n1}
        del s1

    acc_rate = float(sampler.acceptance_rate.mean())
    gather_ms = None
    gather_error = None

    # Outside the metric: the same workload in the other arithmetic mode ('fma'
    # contracts each multiply-add; within 1e-10 of 'exact', not bit-identical).
    other = None
    if not multi and not args.no_other_mode:
        om = 'fma' if args.mode == 'exact' else 'exact'
        s2 = make_sampler(om)
        K2 = min(K, 8)
        W2 = 40 if args.settle_ms > 0 else 2     # another arithmetic mix: settle again
        run(s2, 0, W2)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        run(s2, W2, K2)
        e1.record()
        torch.cuda.synchronize()
        t2 = e0.elapsed_time(e1) * 1e-3 / (K2 * F)
        other = {'mode': om, 'value': C * L / t2, 'avg_transition_us': t2 * 1e6,
                 'roofline_frac': (24.0 * D + 25.0) * C / t2 / 1e9 / HBM_PEAK_GBS,
                 'steps': K2, 'note': 'device time (HIP events), not the headline value'}
        del s2
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64,
                            device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0])
        # the only collective of the path: gather one recorded draw (RCCL); a backend that refuses
        # it costs the two gather figures, not the line
        gather_ms = gather_dst0_ms = None
        try:
            from binf_amd.dist import gather_chains
            state = sampler.state if backend == 'nccl' else sampler.state.cpu()
            gather_chains(state)
            barrier()
            t1 = time.perf_counter()
            for _ in range(5):
                gather_chains(state)
            barrier()
            gather_ms = (time.perf_counter() - t1) / 5 * 1e3
            # gather to rank 0 only (what writing the samples out needs)
            gather_chains(state, dst=0)
            barrier()
            t1 = time.perf_counter()
            for _ in range(5):
                gather_chains(state, dst=0)
            barrier()
            gather_dst0_ms = (time.perf_counter() - t1) / 5 * 1e3
        except Exception as e:                      # noqa: BLE001
            gather_error = '%s: %s' % (type(e).__name__, e)
        else:
            gather_error = None
        # what every rank saw: the line is self-checking for the driver's first SCALE run
        mine = {'rank': rank, 'world_size_seen': dist.get_world_size(), 'backend': dist.get_backend(),
                'device': '%s:%d' % (torch.cuda.get_device_name(dev_index), dev_index),
                'elapsed_s': my_elapsed, 'dev_ms': dev_ms,
                'chain_offset': chain_offset, 'chains': C,
                'value': C * L * K * F / my_elapsed,
                'sustained_value': None if sustained is None else sustained['value_per_gpu']}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    res = None
    if rank == 0:
        transitions = K * F
        steps_total = float(C_total) * L * transitions
        value = steps_total / elapsed
        launch_s = dev_ms * 1e-3 / K                      # per kernel launch (= step)
        trans_s = launch_s / F
        # compulsory bytes of ONE sample() per chain = q0 in +
        # p0 in + q_out out + u, flags, energies = 24 D + 25; one launch = F of them
        contract_bytes_launch = (24.0 * D + 25.0) * C * F
        achieved = contract_bytes_launch / launch_s / 1e9
        # what this launch shape really has to move: p0 in, every thin-th state
        # out, u + flag per transition; q0 in and q_out out once per launch
        moved_bytes_launch = ((8.0 * D * (1.0 + 1.0 / thin) + 9.0) * F + 16.0 * D) * C \
            if F > 1 else contract_bytes_launch
        if live_traffic is not None:
            traffic, traffic_src = live_traffic, live_src
        else:
            traffic, traffic_src = pmc_traffic_committed(C, D, L, F, thin if F > 1 else 1,
                                                         args.mode)
            if traffic_src is not None:
                traffic_src += ' [live PMC pass: %s]' % live_src
        # FP64 lane-operations of one transition: per element 4 L + 2 for the
        # integrator (hmc.py:116-123, gradient = identity for k=1, x0=0) + 6 for
        # the three energy sums; nothing contractible in exact mode
        laneops_launch = (4.0 * L + 2.0 + 6.0) * C * D * F
        if args.mode == 'fma':
            laneops_launch = (2.0 * L + 1.0 + 3.0) * C * D * F
        hbm_meas = (traffic / launch_s / 1e9 / HBM_PEAK_GBS) if traffic is not None else None
        valu_frac = laneops_launch / launch_s / VALU_PEAK_LANEOPS
        roof = {'reading': {
                    'contract_frac': achieved / HBM_PEAK_GBS,
                    'hbm_frac_measured': hbm_meas,
                    'fp64_valu_frac': valu_frac,
                    'single_sample_call_frac': None if single is None else single['frac'],
                    # the other arithmetic mode beside the headline's (other_mode below): 'fma' = each
                    # leapfrog update one fused multiply-add, bit for bit the fused C oracle, within
                    # north_star's 1e-10 of 'exact' with the same golden accept flags
                    'other_mode': None if other is None else other['mode'],
                    'other_mode_contract_frac': None if other is None else other['roofline_frac'],
                    'in_one_sentence': 'north_star asks for >= 0.60 of the HBM roofline: met by the '
                                       'contract reading only (SURVEY 8(d) prices every transition at the '
                                       'bytes of a stand-alone sample()); the persistent kernel keeps q in '
                                       'registers, the bytes that really move are hbm_frac_measured of '
                                       '8 TB/s, and what binds the kernel is the FP64 VALU pipe'},
                'bound': 'hbm', 'achieved': achieved,
                'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS,
                'hbm_frac_moved_of_copy_ceiling': moved_bytes_launch / launch_s / 1e9 / HBM_COPY_GBS,
                'traffic': traffic, 'traffic_source': traffic_src,
                'hbm_frac_measured': hbm_meas,
                'hbm_frac_moved': moved_bytes_launch / launch_s / 1e9 / HBM_PEAK_GBS,
                'valu_frac': valu_frac,
                'single_sample_call': single,
                'per': 'kernel launch = 1 step = %d transition(s), each one '
                       'HMCSampler.sample() worth of work' % F,
                'kernel': PERSIST_KERNEL,
                'transitions_per_launch': F,
                'launches_timed': K,
                'states_recorded': 'every transition' if thin == 1
                else 'every %d. transition' % thin,
                'algorithmic_bytes_per_launch': contract_bytes_launch,
                'algorithmic_contract_bytes': 'SURVEY.md 8(d): (24 D + 25) B per chain and '
                                              'sample() x %d chains x %d transitions' % (C, F),
                'moved_bytes_per_launch': moved_bytes_launch,
                'fp64_lane_ops_per_launch': laneops_launch,
                'valu_peak_lane_ops_per_s': VALU_PEAK_LANEOPS,
                'avg_launch_us': launch_s * 1e6,
                'avg_transition_us': trans_s * 1e6,
                'launch_us_min_median_max': None if per_launch_us is None else
                [per_launch_us[0], per_launch_us[len(per_launch_us) // 2], per_launch_us[-1]],
                'note': 'frac follows the contract (algorithmic bytes of F sample() calls / '
                        'launch time / 8 TB/s); the persistent kernel keeps q in registers, so '
                        'the bytes it really moves are moved_bytes_per_launch (hbm_frac_moved) '
                        'and the PMC-measured traffic (hbm_frac_measured); valu_frac = exact '
                        'FP64 lane-operations / time / 39.3e12 (the kernel is FP64-VALU / '
                        'power bound, DESIGN.md 4.1)'}
        res = {
            'metric': 'chain*leapfrog-steps/sec, 1024-d Gaussian',
            'value': value,
            'unit': 'chain*leapfrog-steps/s',
            'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': elapsed / K * 1e3,
            'higher_is_better': True,
            'scaling': args.scaling,
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': 'C2: %d-d isotropic Gaussian (k=1, x0=0), '
                                   '%s, %d leapfrog steps, dt=%g, '
                                   'fused HMC transition, mode=%s; 1 step = 1 launch = %d '
                                   'transition(s) (sample() calls), %s state recorded; q0 and '
                                   'draw buffer(s) 0..%d from the SURVEY.md 8(d) seeds (host), '
                                   'the other %d filled by torch.randn on the device'
                                   % (D, ('%d chains/GPU' % C) if args.scaling == 'weak' else
                                      ('%d chains in all, %d on rank 0 (strong scaling)'
                                       % (C_total, C)), L, dt, args.mode, F,
                                      'every' if thin == 1 else 'every %d.' % thin,
                                      n_survey - 1, NB - n_survey),
                       'chains_per_gpu': C, 'chains_total': C_total,
                       'n_dims': D, 'leapfrog_steps': L,
                       'transitions_per_step': F,
                       'parallelism': 'chains sharded x%d, no data-path '
                                      'collective' % world,
                       'draw_buffers': NB},
            'timed_region_ms': elapsed * 1e3,
            'settle': {'ms': args.settle_ms, 'launches': n_settle,
                       'first_20_launch_us': cold_us,
                       'what': 'untimed pre-run of the same launches before the %d warm-up steps '
                               '(clock / power controller of the chip settles in ~50 ms of load; '
                               '--settle-ms 0 to disable)' % W},
            'acceptance_rate': acc_rate,
            'roofline': roof,
        }
        if multi:
            res['sample_gather_ms'] = gather_ms
            res['sample_gather_to_rank0_ms'] = gather_dst0_ms
            res['sample_gather_bytes_per_rank'] = C * D * 8
            if gather_error is not None:
                res['sample_gather_error'] = gather_error
            res['ranks'] = per_rank
        if sustained is not None:
            res['value_sustained'] = sustained['value_per_gpu'] * world if not multi else \
                sum(r['sustained_value'] for r in per_rank)
            res['sustained'] = {'launches': sustained['launches'], 'ms': sustained['ms'],
                                'what': 'the same launches for >= %g ms right after the timed steps '
                                        '(device time, HIP events; rank 0 shown, value_sustained sums '
                                        'the ranks)' % args.sustain_ms}
        if other is not None:
            res['other_mode'] = other

    # The sharded C4 / C5 legs (every rank: they hold barriers and the sample gather), and --
    # in a weak-scaling run -- the C2 job of FIXED size (--chains in all) sharded over the same
    # ranks: SURVEY 8(e)'s secondary, strong-scaling figure from the same invocation.
    legs = None
    if multi and not args.no_legs and not args.no_extra:
        del p_bufs, u_bufs, rec_bufs
        torch.cuda.empty_cache()
        from scripts import bench_legs
        comm = bench_legs.Comm(dist, backend, dev)
        guard = LineGuard(res, args.legs_timeout)
        try:
            legs = bench_legs.run_legs(dev, comm, scaling=args.scaling)
        except Exception as e:                      # noqa: BLE001 -- never breaks the headline
            legs = {'error': 'run_legs: %s: %s' % (type(e).__name__, e)}
        if args.scaling == 'weak' and F > 1 and 'error' not in legs:
            try:
                legs['C2_strong'] = c2_strong_leg(args, comm, dev, F, thin, D, L)
            except Exception as e:                  # noqa: BLE001 -- never breaks the headline
                legs['C2_strong'] = {'error': '%s: %s' % (type(e).__name__, e)}
        guard.finish()


    if rank == 0:
        if not multi and not args.no_extra:
            # free the C2 buffers first (the sub-results allocate their own)
            del p_bufs, u_bufs, rec_bufs, sampler
            torch.cuda.empty_cache()
            try:
                from scripts import bench_extra
                res['extra'] = bench_extra.run_all(dev, kstats)
                res['extra']['roofline_kernel_trace'] = kstats_src
            except Exception as e:              # sub-results never break the headline
                res['extra'] = {'error': '%s: %s' % (type(e).__name__, e)}
        if legs is not None:
            # the polynomial (C4) and pair-distance (C5) legs of this N-GPU run, each with
            # its own sample gather and per-rank figures (scripts/bench_legs.py)
            res['extra'] = legs
        if not multi and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(D, L, dt, args.cpu_chains,
                                               args.cpu_calls)
        emit(res)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

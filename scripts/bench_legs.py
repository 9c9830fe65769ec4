"""The sharded legs of ``bench.py --gpus N``: BASELINE configs C4 (Gibbs-within-HMC
on the polynomial model, 32768 chains over 8 GPUs) and C5 (pair-distance
restraint posterior, 3 x 256 coordinates, 2048 chains over 8 GPUs), run on
every rank beside the C2 headline so that one SCALE run of the driver returns
every figure north_star names -- "chain*leapfrog-steps/sec on synthetic
Gaussian and polynomial posteriors at 1, 2, 4 and 8 GPUs" and the RCCL sample
gather of C4 / C5.

One leg = the reference's sampling loop (``example_script.py:33-41``:
``for i in range(n): samples.append(deepcopy(gips.sample()))``, every 20th kept)
on this rank's contiguous block of chains:

* start state and generators are functions of the GLOBAL chain index
  (``DeviceRNG.for_shard``), so the N-GPU job is the one-GPU job, sharded;
* no collective inside the timed sweeps (chains never interact);
* every ``thin``-th state goes to a ``SampleStore`` in HBM (inside the timed
  region -- it is part of producing samples), and the store is gathered to
  rank 0 afterwards, timed on its own: the path's only exchange step
  (``binf/samplers/gibbs.py:136-151`` has none; the reference keeps a Python list).

``value`` of a leg = chains of ALL ranks x L x sweeps / MAX over ranks of the
wall time between two barriers.  Each rank's own figures travel with it
(``ranks``), so the line can be checked against itself: the per-rank values
must add up to ~``value``, ``world_size_seen`` must be N on every rank and the
``chain_offset``s must tile ``[0, C_total)``.

The control flow (:class:`Comm`, :func:`run_leg`) needs no GPU: the CPU suite
drives it with two ``gloo`` ranks and a stand-in leg (``tests/test_bench_launcher.py``).
"""
import time

import numpy as np
import torch


class Comm(object):
    """What a leg needs from ``torch.distributed`` -- or from nothing at all
    when there is one rank.  ``backend`` 'nccl' (= RCCL on ROCm; tensors stay in
    HBM) or 'gloo' (rehearsal: ranks may share a device, collectives move host
    copies)."""

    def __init__(self, dist=None, backend=None, device=None):
        self.dist = dist
        self.backend = backend
        self.device = device
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1

    def sync(self):
        if self.device is not None and torch.device(self.device).type == 'cuda':
            torch.cuda.synchronize()

    def barrier(self):
        self.sync()
        if self.dist is not None:
            self.dist.barrier()
        self.sync()

    def wire_device(self):
        """Where a tensor must live to go through this backend's collectives."""
        return self.device if self.backend == 'nccl' else torch.device('cpu')

    def max(self, x):
        if self.dist is None:
            return float(x)
        t = torch.tensor([float(x)], dtype=torch.float64, device=self.wire_device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def all_gather_object(self, obj):
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out


def run_leg(leg, comm, sweeps, warm=3, thin=5, settle_s=0.1, gather_reps=3):
    """Warm up, then time ``sweeps`` sweeps of ``leg`` between two barriers with
    every ``thin``-th state recorded; then time the gather of the recorded
    draws to rank 0.  Returns the leg's block on every rank (rank 0 prints it).

    ``leg`` provides ``sweep()``, ``state_parts()`` (the tensors of one recorded
    draw, ``[C_local x d_i]`` each), ``n_chains_local``, ``n_chains_total``,
    ``chain_offset``, ``record_width``, ``leapfrog_steps``, ``device`` and
    ``describe()``."""
    from binf_amd.dist import SampleStore

    on_gpu = torch.device(leg.device).type == 'cuda'
    for _ in range(warm):
        leg.sweep()
    comm.sync()
    t_s = time.perf_counter()
    while time.perf_counter() - t_s < settle_s:
        leg.sweep()
        comm.sync()
    n_keep = (sweeps + thin - 1) // thin
    store = SampleStore(n_keep, leg.n_chains_local, leg.record_width, thin=thin,
                        device=leg.device)
    comm.barrier()
    if on_gpu:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    t0 = time.perf_counter()
    for _ in range(sweeps):
        leg.sweep()
        store.record(leg.state_parts())
    if on_gpu:
        e1.record()
    comm.barrier()
    mine = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1) if on_gpu else None
    elapsed = comm.max(mine)

    # the exchange step: the recorded draws to rank 0, timed beside the metric
    gather = None
    if comm.dist is not None:
        wire = store if comm.backend == 'nccl' else store.to('cpu')
        got = wire.gather(leg.n_chains_total, dst=0)
        shape_ok = (got is None) if comm.rank != 0 else \
            (tuple(got.shape) == (store.n_kept, leg.n_chains_total, leg.record_width))
        if comm.rank == 0 and shape_ok:
            # rank 0's own block must sit at its offset, bit for bit
            mine_rows = got[:, leg.chain_offset:leg.chain_offset + leg.n_chains_local]
            shape_ok = torch.equal(mine_rows.cpu(), store.local().cpu())
            check = getattr(leg, 'check_gathered', None)
            if check is not None:
                shape_ok = shape_ok and bool(check(got, thin))
        comm.barrier()
        t1 = time.perf_counter()
        for _ in range(gather_reps):
            wire.gather(leg.n_chains_total, dst=0)
        comm.barrier()
        g_s = (time.perf_counter() - t1) / gather_reps
        per_draw = leg.n_chains_local * leg.record_width * 8
        gather = {'to_rank0_ms': g_s * 1e3,
                  'draws': store.n_kept,
                  'ms_per_recorded_draw': g_s * 1e3 / max(1, store.n_kept),
                  'bytes_per_rank_per_draw': per_draw,
                  'bytes_per_rank': per_draw * store.n_kept,
                  'received_GBps_rank0': per_draw * store.n_kept * (comm.world - 1) / g_s / 1e9,
                  'checked_on_rank0': bool(shape_ok),
                  'collective': 'torch.distributed.gather(dst=0) of the thinned SampleStore, '
                                'backend %s%s' % (comm.backend, '' if comm.backend == 'nccl' else
                                                  ' (host copies: a rehearsal, not RCCL)')}
    L = leg.leapfrog_steps
    ranks = comm.all_gather_object({
        'rank': comm.rank, 'world_size_seen': comm.world,
        'chain_offset': leg.chain_offset, 'chains': leg.n_chains_local,
        'elapsed_s': mine, 'dev_ms': dev_ms,
        'chain_leapfrog_steps_per_s': leg.n_chains_local * L * sweeps / mine,
        'draws_kept': store.n_kept, 'self_check': leg.self_check()})
    tiles, pos = True, 0
    for r in ranks:
        tiles = tiles and r['chain_offset'] == pos
        pos += r['chains']
    out = dict(leg.describe())
    out.update({
        'chain_leapfrog_steps_per_s': leg.n_chains_total * L * sweeps / elapsed,
        'sweep_ms': elapsed / sweeps * 1e3,
        'sweeps_timed': sweeps, 'warmup_sweeps': warm, 'record_every': thin,
        'chains_total': leg.n_chains_total, 'n_gpus': comm.world,
        'sum_of_rank_values': sum(r['chain_leapfrog_steps_per_s'] for r in ranks),
        'shards_tile_the_chains': bool(tiles and pos == leg.n_chains_total),
        'sample_gather': gather,
        'ranks': ranks,
        'timing': 'wall clock between two barriers (+ device synchronise), MAX over ranks; '
                  'the recording of every %d. state is inside, the gather outside' % thin})
    return out


# ---------------------------------------------------------------------------
# the legs
# ---------------------------------------------------------------------------
class _Leg(object):
    leapfrog_steps = 20

    def _shard(self, seed, total, comm, dev):
        from binf_amd.samplers.rng import DeviceRNG
        return DeviceRNG.for_shard(seed, total, comm.rank, comm.world, device=dev)

    def self_check(self):
        return None


class C4Leg(_Leg):
    """C4: Gibbs-within-HMC on the polynomial model (K = 33 coefficients, N = 16384
    data points): HMC (L = 20) on the coefficients + the conjugate
    Gamma draw of the precision per sweep, through the class stack
    (GibbsSampler -> HMCSampler / GammaSampler -> Posterior -> Likelihood)."""

    def __init__(self, dev, comm, chains_per_gpu=4096, K=33, N=16384, L=20, scaling='weak',
                 chains_total=None):
        from binf_amd.dist import shard_chains
        from binf_amd.example.likelihood import POLYVAL, make_likelihood
        from binf_amd.example.priors import GammaPrior, GaussianPrior
        from binf_amd.example.samplers import make_hmc_sampler
        from binf_amd.pdf.posteriors import Posterior
        from binf_amd.samplers import BinfState
        total = chains_total if chains_total is not None else \
            (chains_per_gpu * comm.world if scaling == 'weak' else chains_per_gpu)
        start, count = shard_chains(total, comm.rank, comm.world)
        xs = np.linspace(-1, 1, N)
        c_true = np.random.RandomState(7).standard_normal(K)
        ys = POLYVAL(xs, c_true) + np.random.RandomState(9).standard_normal(N) / np.sqrt(2.5)
        # the whole job's start state, this rank's rows of it
        q0 = torch.from_numpy(np.ascontiguousarray(
            np.random.RandomState(8).standard_normal((total, K))[start:start + count])).to(dev)
        lik = make_likelihood(xs, ys, POLYVAL)
        post = Posterior({lik.name: lik},
                         {'precision_prior': GammaPrior(1.0, 0.2),
                          'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K) * 5)})
        state = BinfState(dict(coefficients=q0,
                               precision=torch.full((count,), 2.5, dtype=torch.float64, device=dev)))
        rng, s1, c1 = self._shard(2, total, comm, dev)
        grng, _, _ = self._shard(5, total, comm, dev)
        assert (s1, c1) == (start, count)
        self.gips = make_hmc_sampler(post, 2e-4, L, state, rng=rng, gamma=grng.gamma)
        self.device = dev
        self.n_chains_local, self.n_chains_total, self.chain_offset = count, total, start
        self.record_width = K + 1
        self.leapfrog_steps = L
        self._what = (K, N, L, scaling)

    def sweep(self):
        self.gips.sample()

    def state_parts(self):
        v = self.gips.state.variables
        return (v['coefficients'], v['precision'])

    def self_check(self):
        v = self.gips.state.variables
        hmc = self.gips.subsamplers['coefficients']
        return {'acceptance': float(hmc.acceptance_rate.mean()),
                'precision_mean': float(v['precision'].mean()),
                'finite': bool(torch.isfinite(v['coefficients']).all())}

    def describe(self):
        K, N, L, scaling = self._what
        return {'workload': 'C4: Gibbs-within-HMC, polynomial K=%d, N=%d, %d chains in all (%s '
                            'scaling), L=%d; one sweep = HMC on the coefficients + Gamma draw of '
                            'the precision; a recorded draw = coefficients + precision = %d B '
                            'per chain' % (K, N, self.n_chains_total, scaling, L, 8 * (K + 1)),
                'unit': 'chain*leapfrog-steps/s'}


class C5Leg(_Leg):
    """C5: the pair-distance restraint posterior, 256 beads x 3 coordinates
    (build-defined; the reference names the application only), HMC (L = 20) on the
    coordinates with the precision fixed, fused leapfrog + one-launch energy."""

    def __init__(self, dev, comm, chains_per_gpu=256, n=256, L=20, scaling='weak',
                 chains_total=None):
        from binf_amd.dist import shard_chains
        from binf_amd.example.distance import make_distance_likelihood
        from binf_amd.pdf import IsotropicGaussian
        from binf_amd.pdf.posteriors import Posterior
        from binf_amd.samplers.hmc import HMCSampler
        total = chains_total if chains_total is not None else \
            (chains_per_gpu * comm.world if scaling == 'weak' else chains_per_gpu)
        start, count = shard_chains(total, comm.rank, comm.world)
        rs = np.random.RandomState(0)
        truth = rs.standard_normal((n, 3)) * 2.0
        I, J = np.triu_indices(n, 1)
        d_true = np.sqrt(np.sum((truth[I] - truth[J]) ** 2, axis=1))
        ys = np.abs(d_true + 0.05 * rs.standard_normal(n * (n - 1) // 2))
        x = torch.from_numpy(np.ascontiguousarray(
            (truth.reshape(-1)[None, :] +
             0.1 * np.random.RandomState(1).standard_normal((total, 3 * n)))[start:start + count])).to(dev)
        lik = make_distance_likelihood(ys, n)
        prior = IsotropicGaussian(0.05, 0.0, name='coordinates_prior', variable_name='coordinates')
        cond = Posterior({lik.name: lik}, {prior.name: prior}).conditional_factory(precision=4.0)
        rng, s1, c1 = self._shard(0, total, comm, dev)
        assert (s1, c1) == (start, count)
        self.sampler = HMCSampler(cond, x, 0.002, L, variable_name='coordinates', rng=rng)
        self.device = dev
        self.n_chains_local, self.n_chains_total, self.chain_offset = count, total, start
        self.record_width = 3 * n
        self.leapfrog_steps = L
        self._what = (n, L, scaling)

    def sweep(self):
        self.sampler.sample()

    def state_parts(self):
        return self.sampler.state

    def self_check(self):
        return {'acceptance': float(self.sampler.acceptance_rate.mean()),
                'finite': bool(torch.isfinite(self.sampler.state).all())}

    def describe(self):
        n, L, scaling = self._what
        return {'workload': 'C5: pair-distance restraint posterior, %d beads x 3, %d chains in '
                            'all (%s scaling), L=%d; one sweep = one HMCSampler.sample(); a '
                            'recorded draw = %d B per chain'
                            % (n, self.n_chains_total, scaling, L, 8 * 3 * n),
                'unit': 'chain*leapfrog-steps/s'}


class StandInLeg(_Leg):
    """The control flow without a GPU (BINF_BENCH_DRYRUN, the CPU suite): a
    "sweep" adds one to a host tensor whose rows carry their global chain index,
    so the gathered store can be checked for order and content."""

    def __init__(self, comm, chains_per_gpu=6, width=3, scaling='weak'):
        from binf_amd.dist import shard_chains
        total = chains_per_gpu * comm.world if scaling == 'weak' else chains_per_gpu
        start, count = shard_chains(total, comm.rank, comm.world)
        self.device = torch.device('cpu')
        self.n_chains_local, self.n_chains_total, self.chain_offset = count, total, start
        self.record_width = width
        self.leapfrog_steps = 20
        self.x = torch.arange(start, start + count, dtype=torch.float64).reshape(-1, 1) * \
            torch.ones((1, width - 1), dtype=torch.float64)
        self.t = torch.zeros(count, dtype=torch.float64)
        self.scaling = scaling

    def sweep(self):
        self.t = self.t + 1.0

    def state_parts(self):
        return (self.x, self.t)

    def self_check(self):
        return {'sweeps_seen': float(self.t[0]) if self.t.numel() else None}

    def check_gathered(self, got, thin):
        """Rows in global chain order; draw k is the state after sweep 1 + k * thin of the
        timed loop (which followed the warm-up sweeps: the first kept value says how many)."""
        w = self.record_width
        rows = torch.arange(self.n_chains_total, dtype=torch.float64)
        ok = all(torch.equal(got[k, :, j], rows) for k in range(got.shape[0]) for j in range(w - 1))
        t0 = float(got[0, 0, w - 1])
        return ok and all(torch.equal(got[k, :, w - 1], torch.full_like(rows, t0 + k * thin))
                          for k in range(got.shape[0]))

    def describe(self):
        return {'workload': 'stand-in leg (dry run, %s scaling)' % self.scaling,
                'unit': 'chain*leapfrog-steps/s'}


def run_legs(dev, comm, scaling='weak', c4_sweeps=20, c5_sweeps=100):
    """Both legs on this rank's shard; ``{'C4': ..., 'C5': ...}`` (a failing leg
    becomes an ``error`` entry on every rank -- the collectives of the other leg
    still line up because the failure is agreed on first)."""
    out = {}
    for name, make, sweeps, thin in (
            ('C4', lambda: C4Leg(dev, comm, scaling=scaling,
                                 chains_per_gpu=4096 if scaling == 'weak' else 32768), c4_sweeps, 5),
            ('C5', lambda: C5Leg(dev, comm, scaling=scaling,
                                 chains_per_gpu=256 if scaling == 'weak' else 2048), c5_sweeps, 20)):
        err = None
        leg = None
        try:
            leg = make()
            leg.sweep()                       # a first sweep before anyone commits to the leg
            comm.sync()
        except Exception as e:                # noqa: BLE001 -- reported, never hides the headline
            err = '%s: %s' % (type(e).__name__, e)
        errs = [e for e in comm.all_gather_object(err) if e is not None]
        if errs:
            out[name] = {'error': errs[0], 'ranks_failed': len(errs)}
        else:
            try:
                out[name] = run_leg(leg, comm, sweeps, thin=thin)
            except Exception as e:            # noqa: BLE001 -- e.g. a collective refused by the backend:
                out[name] = {'error': 'run_leg: %s: %s' % (type(e).__name__, e)}    # the headline still prints
        del leg
        if torch.device(dev).type == 'cuda':
            torch.cuda.empty_cache()
    return out

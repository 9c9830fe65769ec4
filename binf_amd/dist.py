"""
Chains across GPUs.

Chains never interact (the reference has no cross-chain operation), so the
path shards trivially: one process per GPU, each owning a contiguous block of
chains, NO collective while sampling.  The only exchange is gathering drawn
samples -- ``torch.distributed`` with the ``nccl`` backend, which is RCCL over
xGMI on ROCm (``gloo`` on CPU, used by the tests).  Because an all-gather of
every draw would cost more than producing it (32 MiB per GPU per draw at C2),
draws are recorded into an on-device store, thinned (the reference's own
script keeps 1 in 20 after burn-in, ``example_script.py:41``), and gathered
once.
"""
import os

import torch


def _dist():
    import torch.distributed as dist
    return dist


def world():
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_chains(n_chains, rank=None, world_size=None):
    """Contiguous block of chains owned by ``rank``: ``(start, count)``.  The
    first ``n_chains % world_size`` ranks own one chain more."""
    if rank is None or world_size is None:
        rank, world_size = world()
    if n_chains < 0 or world_size < 1 or not 0 <= rank < world_size:
        raise ValueError('bad shard request: n_chains=%r rank=%r world=%r'
                         % (n_chains, rank, world_size))
    base, extra = divmod(n_chains, world_size)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


class PendingGather(object):
    """An all-gather in flight (``gather_chains(..., async_op=True)``): the
    collective runs on the backend's own stream (RCCL: beside the sampling
    kernels on xGMI), ``wait()`` returns the gathered tensor."""

    def __init__(self, work, out, finish):
        self._work, self._out, self._finish = work, out, finish

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._finish(self._out)


def gather_chains(local, n_chains_total=None, group=None, async_op=False, dst=None):
    """Gather a per-chain tensor along dim 0 (chains).  ``local`` is
    ``[C_local, ...]``; returns ``[C_total, ...]``, rows in global chain order.
    Uneven shards (see :func:`shard_chains`) are padded to the largest shard for
    the collective and trimmed afterwards.

    ``dst=None``: all-gather, every rank gets the result.  ``dst=r``: a gather
    to rank ``r`` only (the usual case: one rank writes the samples out) --
    the other ranks send their shard once and get ``None``; each rank then
    moves ``1 / world_size`` of the all-gather's receive traffic.

    ``async_op=True`` returns a :class:`PendingGather` at once: sampling can go on
    while the collective runs (do not overwrite ``local`` before ``wait()``)."""
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()) or \
            (dist.get_world_size(group) == 1 and os.environ.get('BINF_DIST_NO_SHORTCUT') != '1'):
        # (BINF_DIST_NO_SHORTCUT=1: a one-rank group still goes through the collective -- how a
        # one-GPU box rehearses the RCCL calls themselves, bench.py BINF_BENCH_FORCE_DIST)
        return PendingGather(None, local, lambda t: t) if async_op else local
    ws = dist.get_world_size(group)
    if dst is not None and not 0 <= int(dst) < ws:
        raise ValueError('gather_chains: dst=%r outside the group of %d ranks' % (dst, ws))
    c_local = local.shape[0]
    if n_chains_total is None:
        counts = [c_local] * ws
        same = torch.tensor([c_local], dtype=torch.int64, device=local.device)
        lo, hi = same.clone(), same.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        if int(lo) != int(hi):
            raise ValueError('uneven shards need n_chains_total')
    else:
        counts = [shard_chains(n_chains_total, r, ws)[1] for r in range(ws)]
        if counts[dist.get_rank(group)] != c_local:
            raise ValueError('local shard has %d chains, expected %d'
                             % (c_local, counts[dist.get_rank(group)]))
    cmax = max(counts)
    send = local.contiguous()
    if c_local < cmax:
        pad = torch.zeros((cmax - c_local,) + tuple(local.shape[1:]),
                          dtype=local.dtype, device=local.device)
        send = torch.cat([send, pad], dim=0)
    def finish(t):
        if t is None or all(c == cmax for c in counts):
            return t
        return torch.cat([t[r * cmax:r * cmax + counts[r]] for r in range(ws)], dim=0)
    if dst is not None:
        me = dist.get_rank(group)
        out, parts = None, None
        if me == int(dst):
            out = torch.empty((ws * cmax,) + tuple(local.shape[1:]), dtype=local.dtype,
                              device=local.device)
            parts = list(out.view((ws, cmax) + tuple(local.shape[1:])).unbind(0))
        gdst = dist.get_global_rank(group, int(dst)) if group is not None else int(dst)
        work = dist.gather(send, parts, dst=gdst, group=group, async_op=async_op)
        if async_op:
            return PendingGather(work, out, finish)
        return finish(out)
    out = torch.empty((ws * cmax,) + tuple(local.shape[1:]), dtype=local.dtype,
                      device=local.device)
    if async_op:
        return PendingGather(dist.all_gather_into_tensor(out, send, group=group, async_op=True),
                             out, finish)
    dist.all_gather_into_tensor(out, send, group=group)
    return finish(out)


class SampleStore(object):
    """Thinned on-device record of drawn samples with one deferred gather.

    ``record(x)`` is called after every ``sample()``; every ``thin``-th call
    (after ``burn_in`` calls) copies the ``[C_local x D]`` state into a
    preallocated ``[capacity, C_local, D]`` buffer in HBM -- replaces the
    Python list of deep-copied states of ``example_script.py:32-34``.
    ``gather()`` returns ``[n_kept, C_total, D]`` on every rank.
    """

    def __init__(self, capacity, n_chains_local, n_dims, thin=1, burn_in=0,
                 device=None, dtype=torch.float64):
        if capacity < 1 or thin < 1 or burn_in < 0:
            raise ValueError('capacity >= 1, thin >= 1, burn_in >= 0 required')
        self.thin = thin
        self.burn_in = burn_in
        self.buffer = torch.empty((capacity, n_chains_local, n_dims),
                                  dtype=dtype, device=device)
        self.n_seen = 0
        self.n_kept = 0

    def record(self, x):
        """Returns True if the draw was kept.  ``x`` is the ``[C_local x D]`` state, or
        a tuple of per-chain tensors (``[C_local x d_i]`` / ``[C_local]``) whose widths
        add up to D -- the variables of a Gibbs state, laid side by side in the slot
        (coefficients ``[C x K]`` + precision ``[C]`` -> ``[C x (K + 1)]``) without an
        intermediate concatenation."""
        i = self.n_seen
        self.n_seen += 1
        if i < self.burn_in or (i - self.burn_in) % self.thin != 0:
            return False
        if self.n_kept >= self.buffer.shape[0]:
            raise IndexError('SampleStore is full (%d draws)' % self.n_kept)
        slot = self.buffer[self.n_kept]
        if isinstance(x, (tuple, list)):
            col = 0
            for part in x:
                part = part.reshape(slot.shape[0], -1)
                slot[:, col:col + part.shape[1]].copy_(part)
                col += part.shape[1]
            if col != slot.shape[1]:
                raise ValueError('SampleStore.record: parts are %d wide, the store %d'
                                 % (col, slot.shape[1]))
        else:
            slot.copy_(x.reshape(slot.shape))
        self.n_kept += 1
        return True

    # -- checkpoint / resume (binf_amd/checkpoint.py) ----------------------------------
    def state_dict(self):
        return {'kept': self.buffer[:self.n_kept], 'n_seen': int(self.n_seen), 'n_kept': int(self.n_kept),
                'thin': int(self.thin), 'burn_in': int(self.burn_in)}

    def load_state_dict(self, d):
        if int(d['thin']) != self.thin or int(d['burn_in']) != self.burn_in:
            raise ValueError('SampleStore checkpoint has thin=%s, burn_in=%s' % (d['thin'], d['burn_in']))
        n = int(d['n_kept'])
        if n > self.buffer.shape[0] or tuple(d['kept'].shape[1:]) != tuple(self.buffer.shape[1:]):
            raise ValueError('SampleStore checkpoint of shape %s does not fit a store of %s'
                             % (tuple(d['kept'].shape), tuple(self.buffer.shape)))
        self.buffer[:n].copy_(d['kept'].to(self.buffer.device))
        self.n_seen, self.n_kept = int(d['n_seen']), n

    def to(self, device):
        """A store on ``device`` holding the draws kept so far (a host copy is what
        the ``gloo`` backend can move; RCCL gathers straight from HBM)."""
        other = SampleStore.__new__(SampleStore)
        other.thin, other.burn_in = self.thin, self.burn_in
        other.buffer = self.buffer.to(device)
        other.n_seen, other.n_kept = self.n_seen, self.n_kept
        return other

    def extend(self, block, n_sweeps=None):
        """Append a block of already thinned draws ``[m, C_local, D]`` -- what
        ``GibbsSampler.sample_n`` / ``HMCSampler.sample_n`` return -- that stands
        for ``n_sweeps`` calls of ``record`` (default ``m * thin``)."""
        m = int(block.shape[0])
        if self.n_kept + m > self.buffer.shape[0]:
            raise IndexError('SampleStore is full (%d + %d draws)' % (self.n_kept, m))
        self.buffer[self.n_kept:self.n_kept + m].copy_(
            block.reshape((m,) + tuple(self.buffer.shape[1:])))
        self.n_kept += m
        self.n_seen += m * self.thin if n_sweeps is None else int(n_sweeps)

    def local(self):
        return self.buffer[:self.n_kept]

    def gather(self, n_chains_total=None, group=None, async_op=False, dst=None):
        """``[n_kept, C_total, D]`` on every rank (``dst=r``: on rank ``r`` only,
        ``None`` elsewhere); with ``async_op=True`` a :class:`PendingGather` (the
        store may keep recording into its later slots meanwhile: the draws kept
        so far were copied out for the collective)."""
        kept = self.local()
        back = lambda g: None if g is None else g.transpose(0, 1).contiguous()
        if kept.shape[0] == 0:
            # nothing kept yet: no collective; the contract of ``dst`` still holds
            # (None on every rank but ``dst``)
            rank, ws = world() if group is None else (_dist().get_rank(group),
                                                      _dist().get_world_size(group))
            empty = kept if (dst is None or ws == 1 or rank == int(dst)) else None
            return PendingGather(None, empty, lambda t: t) if async_op else empty
        # chains to dim 0 for the collective, back afterwards
        g = gather_chains(kept.transpose(0, 1).contiguous(), n_chains_total, group,
                          async_op=async_op, dst=dst)
        if async_op:
            inner = g
            return PendingGather(None, None, lambda _t: back(inner.wait()))
        return back(g)

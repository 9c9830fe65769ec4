"""World-size-2 CPU tests (gloo) of the multi-GPU path: chain sharding and the
deferred sample gather.  The sampling itself needs no collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from binf_amd.dist import SampleStore, gather_chains, shard_chains, world


def test_shard_chains_partitions_exactly():
    for n, ws in [(0, 1), (1, 1), (7, 2), (8, 2), (4096, 8), (32768, 8), (5, 8), (1001, 6)]:
        pos = 0
        for r in range(ws):
            start, count = shard_chains(n, r, ws)
            assert start == pos and count >= 0
            pos += count
        assert pos == n
        counts = [shard_chains(n, r, ws)[1] for r in range(ws)]
        assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        shard_chains(4, 3, 2)
    assert world() == (0, 1)


def test_sample_store_thins_and_burns_in_without_a_process_group():
    st = SampleStore(capacity=4, n_chains_local=3, n_dims=2, thin=3, burn_in=2)
    kept = []
    for i in range(12):
        x = torch.full((3, 2), float(i), dtype=torch.float64)
        if st.record(x):
            kept.append(i)
    assert kept == [2, 5, 8, 11]
    assert st.local().shape == (4, 3, 2)
    assert st.gather().shape == (4, 3, 2)           # single process: identity
    assert torch.equal(st.gather(async_op=True).wait(), st.local())
    assert [float(v) for v in st.local()[:, 0, 0]] == [2.0, 5.0, 8.0, 11.0]
    with pytest.raises(IndexError):
        for i in range(3):
            st.record(torch.zeros((3, 2), dtype=torch.float64))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, ws, port, n_total, n_dims, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    try:
        assert world() == (rank, ws)
        start, count = shard_chains(n_total)
        # every chain's row holds its GLOBAL index: the gather must restore order
        glob = torch.arange(n_total, dtype=torch.float64).reshape(-1, 1) * \
            torch.ones((1, n_dims), dtype=torch.float64)
        local = glob[start:start + count].clone()
        full = gather_chains(local, n_total)
        ok = torch.equal(full, glob)
        flags = gather_chains((local[:, 0] % 2 == 0).to(torch.uint8), n_total)
        ok = ok and torch.equal(flags, (glob[:, 0] % 2 == 0).to(torch.uint8))
        # deferred, thinned gather of several draws
        st = SampleStore(capacity=3, n_chains_local=count, n_dims=n_dims, thin=2)
        for i in range(6):
            st.record(local + 1000.0 * i)
        g = st.gather(n_total)
        ok = ok and g.shape == (3, n_total, n_dims)
        for k, i in enumerate((0, 2, 4)):
            ok = ok and torch.equal(g[k], glob + 1000.0 * i)
        # the same gathers in flight while "sampling" goes on (async_op): the store
        # keeps recording into its later slots meanwhile
        st2 = SampleStore(capacity=4, n_chains_local=count, n_dims=n_dims)
        st2.record(local)
        st2.record(local + 1000.0)
        pend = st2.gather(n_total, async_op=True)
        st2.record(local + 2000.0)                      # does not disturb the gather in flight
        pend_flags = gather_chains((local[:, 0] % 2 == 0).to(torch.uint8), n_total, async_op=True)
        g2 = pend.wait()
        ok = ok and g2.shape == (2, n_total, n_dims) and torch.equal(g2[1], glob + 1000.0)
        ok = ok and torch.equal(pend_flags.wait(), (glob[:, 0] % 2 == 0).to(torch.uint8))
        ok = ok and st2.gather(n_total).shape == (3, n_total, n_dims)
        # gather to ONE rank (dst): the others send and get None
        for d in range(ws):
            got = gather_chains(local, n_total, dst=d)
            ok = ok and ((torch.equal(got, glob)) if rank == d else got is None)
        pend = gather_chains(local, n_total, dst=0, async_op=True)
        got = pend.wait()
        ok = ok and (torch.equal(got, glob) if rank == 0 else got is None)
        gd = st.gather(n_total, dst=1)
        ok = ok and ((gd.shape == (3, n_total, n_dims) and torch.equal(gd[2], glob + 4000.0))
                     if rank == 1 else gd is None)
        # the device generator of this rank's shard: same seed, chain_offset = the
        # shard's first GLOBAL chain (what makes an N-GPU run reproduce the 1-GPU run)
        from binf_amd.samplers.rng import DeviceRNG
        rng, s0, c0 = DeviceRNG.for_shard(17, n_total)
        ok = ok and (s0, c0) == (start, count) and rng.chain_offset == start and rng.seed == 17
        offs = [None] * ws
        dist.all_gather_object(offs, (rng.chain_offset, count))
        ok = ok and offs[0][0] == 0 and all(offs[r + 1][0] == offs[r][0] + offs[r][1]
                                             for r in range(ws - 1))
        ok = ok and offs[-1][0] + offs[-1][1] == n_total
        ok = ok and rng._elem_offset((count, n_dims)) == start * n_dims
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and float(t) == float(ws)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_total', [8, 7])
def test_two_rank_shard_and_gather(n_total):
    ws, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, ws, port, n_total, 5, q))
             for r in range(ws)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(ws)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]

"""RCCL smoke on the GPU box: the collectives the multi-GPU path uses
(all_gather_into_tensor of a recorded draw, MAX all-reduce of the timing, barrier)
through torch.distributed's `nccl` backend (= RCCL on ROCm) with the ONE rank a
one-GPU box allows, in this process (no child process: the GPU is initialised).
The N > 1 logic is covered on the CPU with gloo (tests/test_dist_gloo.py,
tests/test_bench_launcher.py); this only establishes that the RCCL path itself
initialises and runs where the driver will scale it."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_collectives(device):
    import torch.distributed as dist
    from binf_amd.dist import SampleStore, gather_chains, shard_chains, world
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    saved = {k: os.environ.get(k) for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR',
                                            'MASTER_PORT')}
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    try:
        torch.cuda.set_device(device)
        dist.init_process_group('nccl', device_id=device)
        try:
            assert world() == (0, 1)
            x = torch.arange(12, dtype=torch.float64, device=device).reshape(4, 3)
            out = torch.empty_like(x)
            dist.all_gather_into_tensor(out, x)          # the collective gather_chains issues
            assert torch.equal(out, x)
            t = torch.tensor([3.5], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)     # bench.py's max-over-ranks
            assert float(t) == 3.5
            dist.barrier()
            assert torch.equal(gather_chains(x, 4), x) and shard_chains(4) == (0, 4)
            st = SampleStore(capacity=2, n_chains_local=4, n_dims=3, device=device)
            st.record(x)
            st.record(x + 1)
            assert st.gather(4).shape == (2, 4, 3)
            # ... and the same calls with the one-rank short cut of binf_amd/dist.py switched
            # off: gather_chains / SampleStore.gather now issue dist.gather(dst=0) and
            # all_gather_into_tensor on RCCL (sync and async), as every rank of an N-GPU run does
            os.environ['BINF_DIST_NO_SHORTCUT'] = '1'
            try:
                assert torch.equal(gather_chains(x, 4), x)
                assert torch.equal(gather_chains(x, 4, dst=0), x)
                assert torch.equal(gather_chains(x, 4, dst=0, async_op=True).wait(), x)
                assert torch.equal(gather_chains(x, 4, async_op=True).wait(), x)
                got = st.gather(4, dst=0)
                assert got.shape == (2, 4, 3) and torch.equal(got[1], x + 1)
                objs = [None]
                dist.all_gather_object(objs, {'rank': 0, 'value': 1.5})   # bench.py's per-rank fields
                assert objs == [{'rank': 0, 'value': 1.5}]
            finally:
                del os.environ['BINF_DIST_NO_SHORTCUT']
        finally:
            dist.destroy_process_group()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

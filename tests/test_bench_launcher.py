"""bench.py starts its own ranks for --gpus N > 1 (CPU suite: the control flow
only -- rendezvous on 127.0.0.1, barrier, MAX over ranks, the sample gather --
through BINF_BENCH_DRYRUN; the sampling itself needs a GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env(**kw):
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(kw)
    return env


def test_self_launch_two_ranks_gloo_dry_run():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--chains', '6'],
                       env=_env(BINF_BENCH_DRYRUN='1', BINF_BENCH_BACKEND='gloo'),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1                       # rank 0 only
    res = json.loads(lines[0])
    assert res['dry_run'] is True and res['n_gpus'] == 2
    assert res['steps'] == 3 and res['warmup'] == 1
    assert res['max_elapsed'] == 2.0             # MAX over ranks of 1 + rank
    assert res['gathered_rows'] == 12            # 6 chains per rank, weak scaling


def test_more_ranks_than_gpus_is_a_clear_error():
    import torch
    ndev = torch.cuda.device_count()
    n = max(2, ndev + 1)
    r = subprocess.run([sys.executable, BENCH, '--gpus', str(n)],
                       env=_env(BINF_BENCH_BACKEND='nccl'),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 2
    err = r.stderr.decode()
    assert 'only %d GPU(s) visible' % ndev in err
    assert 'torch.distributed.run' not in err


def test_single_rank_dry_run_needs_no_process_group():
    r = subprocess.run([sys.executable, BENCH], env=_env(BINF_BENCH_DRYRUN='1'),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    res = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert res['n_gpus'] == 1 and res['steps'] == 20 and res['warmup'] == 5

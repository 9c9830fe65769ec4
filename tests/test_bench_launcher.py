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
    # ... and nothing else on stdout: what libraries print there (gloo's connection lines, RCCL's
    # version banner on the GPU) is sent to stderr (bench.py: claim_stdout)
    assert r.stdout.decode().strip().splitlines() == lines
    res = json.loads(lines[0])
    assert res['dry_run'] is True and res['n_gpus'] == 2
    assert res['steps'] == 3 and res['warmup'] == 1
    assert res['max_elapsed'] == 2.0             # MAX over ranks of 1 + rank
    assert res['gathered_rows'] == 12            # 6 chains per rank, weak scaling
    assert res['scaling'] == 'weak' and res['shard'] == [0, 6]
    _check_leg(res['extra']['stand_in'], world=2, total=12)


def _check_leg(leg, world, total):
    """The block a sharded leg (scripts/bench_legs.py: C4 / C5 on the GPUs, a stand-in
    here) puts under `extra`: per-rank figures that add up, shards that tile the chains,
    the gather of the recorded draws to rank 0 timed and checked."""
    assert leg['n_gpus'] == world and leg['chains_total'] == total
    assert leg['shards_tile_the_chains'] is True
    ranks = leg['ranks']
    assert [r['rank'] for r in ranks] == list(range(world))
    assert all(r['world_size_seen'] == world for r in ranks)
    assert sum(r['chains'] for r in ranks) == total
    pos = 0
    for r in ranks:
        assert r['chain_offset'] == pos
        pos += r['chains']
    # value = all chains x L x sweeps / MAX over ranks of the wall time
    slowest = max(r['elapsed_s'] for r in ranks)
    assert leg['chain_leapfrog_steps_per_s'] <= total * 20 * leg['sweeps_timed'] / slowest * 1.0001
    assert abs(leg['sum_of_rank_values'] - sum(r['chain_leapfrog_steps_per_s'] for r in ranks)) \
        <= 1e-9 * leg['sum_of_rank_values']
    assert leg['sum_of_rank_values'] >= leg['chain_leapfrog_steps_per_s'] * 0.9999
    assert all(r['draws_kept'] == 3 for r in ranks)          # sweeps 0, 2, 4 of 6
    g = leg['sample_gather']
    if world == 1:
        assert g is None
    else:
        assert g['checked_on_rank0'] is True and g['draws'] == 3
        assert g['bytes_per_rank_per_draw'] == ranks[0]['chains'] * 3 * 8
        assert g['to_rank0_ms'] > 0


def test_strong_scaling_shards_the_total_and_uneven_shards_gather():
    """--scaling strong: --chains is the whole job; 7 chains over 2 ranks = 4 + 3 (the
    gather pads the short shard and trims it again)."""
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--chains', '7', '--scaling', 'strong'],
                       env=_env(BINF_BENCH_DRYRUN='1', BINF_BENCH_BACKEND='gloo'),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    res = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith('{')][-1])
    assert res['scaling'] == 'strong' and res['shard'] == [0, 4] and res['gathered_rows'] == 7
    leg = res['extra']['stand_in']
    _check_leg(leg, world=2, total=7)
    assert [r_['chains'] for r_ in leg['ranks']] == [4, 3]


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
    _check_leg(res['extra']['stand_in'], world=1, total=4096)


def test_a_rank_that_never_reaches_the_legs_costs_the_legs_not_the_line():
    """bench.py's LineGuard: rank 1 hangs before the sharded legs (test hook), rank 0 waits
    for it in the leg's first collective -- after --legs-timeout rank 0 prints the line it
    already held, with the reason in place of the legs, and both ranks leave with code 0."""
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--chains', '6', '--legs-timeout', '4'],
                       env=_env(BINF_BENCH_DRYRUN='1', BINF_BENCH_BACKEND='gloo',
                                BINF_BENCH_DRYRUN_HANG_RANK='1'),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['max_elapsed'] == 2.0 and res['gathered_rows'] == 12
    assert 'did not finish within 4 s' in res['extra']['error']

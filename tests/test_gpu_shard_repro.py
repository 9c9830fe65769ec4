"""A sharded run reproduces the unsharded run, bit for bit.

Chains are independent (the reference has no cross-chain term,
``binf/samplers/hmc.py:136-164``), so an N-GPU run owns contiguous blocks of
chains (``dist.shard_chains``).  Every device random stream is keyed by the
GLOBAL chain index (``chain_offset`` through the C ABI): what a chain draws --
and hence its whole trajectory -- does not depend on the batch it sits in.
Checked here on ONE GPU: the full batch vs its shards run one after the other.
"""
import numpy as np
import pytest
import torch

from binf_amd import _native
from binf_amd.pdf import native_gauss
from binf_amd.dist import shard_chains
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG

pytestmark = pytest.mark.gpu


def run(q0, rng, n, L=5, dt=None, single_calls=False, **kw):
    C, D = q0.shape
    dt = dt if dt is not None else 0.9 / np.sqrt(max(D, 4))
    s = HMCSampler(IsotropicGaussian(), q0.clone(), dt, L, variable_name='x', rng=rng,
                   record_energies=True, **kw)
    if single_calls:
        rec, acc, eb, ea = [], [], [], []
        for _ in range(n):
            rec.append(s.sample())
            acc.append(s.last_move_accepted)
            eb.append(s.last_e_before)
            ea.append(s.last_e_after)
        return torch.stack(rec), torch.stack(acc), torch.stack(eb), torch.stack(ea), s
    rec = s.sample_n(n)
    return rec, s.accepted_history, s.last_e_before, s.last_e_after, s


def assert_shards_equal_full(device, C, D, parts, n=3, single_calls=False, fused=True, **kw):
    q0 = torch.from_numpy(np.random.RandomState(C + D).standard_normal((C, D))).to(device)
    seed = 4242
    full = run(q0, DeviceRNG(seed, device, fused=fused), n, single_calls=single_calls, **kw)
    assert float(full[1].double().mean()) > 0
    for r in range(parts):
        rng, start, count = DeviceRNG.for_shard(seed, C, rank=r, world_size=parts, device=device,
                                                fused=fused)
        assert rng.chain_offset == start == shard_chains(C, r, parts)[0]
        if count == 0:
            continue
        part = run(q0[start:start + count].contiguous(), rng, n, single_calls=single_calls, **kw)
        sl = slice(start, start + count)
        assert torch.equal(part[0], full[0][:, sl]), (r, 'samples')
        assert torch.equal(part[1], full[1][:, sl]), (r, 'accept flags')
        assert torch.equal(part[2], full[2][:, sl]) and torch.equal(part[3], full[3][:, sl]), (r, 'energies')
        assert torch.equal(part[4].state, full[4].state[sl])
        assert torch.equal(part[4].n_accepted, full[4].n_accepted[sl])
        if isinstance(full[4].timestep, torch.Tensor):
            assert torch.equal(part[4].timestep, full[4].timestep[sl])


def test_c2_halves_and_eighths_reproduce_the_full_batch(device):
    """BASELINE C2's batch (4096 x 1024): the full launch generates its draws in
    the kernel (one wave per chain); two halves (2048 chains: two waves per
    chain, in-kernel draws) and eight eighths (512 chains: the 4-wave split
    kernel fed by the draw kernel) give the same states, flags and energies."""
    C, D = 4096, 1024
    rng = DeviceRNG(1, device)
    s = HMCSampler(IsotropicGaussian(), torch.zeros((8, D), dtype=torch.float64, device=device),
                   0.02, 3, variable_name='x', rng=rng)
    assert native_gauss.draws_in_kernel(s, 4096, D) and native_gauss.draws_in_kernel(s, 2048, D)
    assert not native_gauss.draws_in_kernel(s, 512, D) and _native.gauss_waves_per_chain(512, D) == 4
    assert_shards_equal_full(device, C, D, 2, n=3, dt=0.05, L=20)
    assert_shards_equal_full(device, C, D, 8, n=2, dt=0.05, L=20)


@pytest.mark.parametrize('C,D,parts', [(70, 768, 3), (37, 33, 4), (10, 200, 7), (9, 2048, 2),
                                       (5, 7000, 2), (300, 7, 3), (3, 1024, 5)])
def test_shards_reproduce_the_full_batch_ragged(device, C, D, parts):
    """Uneven shards (shard_chains gives the first ranks one chain more; a rank
    may own nothing), several chains per wave, chains of several waves, adaption."""
    assert_shards_equal_full(device, C, D, parts, timestep_adaption_limit=3)
    assert_shards_equal_full(device, C, D, parts, single_calls=True, mode='fma')


@pytest.mark.parametrize('C,D,parts', [(6, 8193, 2), (5, 20000, 3)])
def test_long_chain_shards_reproduce_the_full_batch(device, C, D, parts):
    """Chains beyond the persistent kernel (hmc_gauss_big.hip): momentum streams
    per (global chain, chunk, lane), acceptance draw per global chain."""
    assert_shards_equal_full(device, C, D, parts, n=2, single_calls=True, timestep_adaption_limit=4)


@pytest.mark.parametrize('C,D,parts', [(64, 1024, 2), (11, 33, 3), (7, 9, 2)])
def test_philox_shards_reproduce_the_full_batch(device, C, D, parts):
    """fused=False: the stand-alone Philox kernels, element i of a [C_total x D]
    draw a function of its GLOBAL flat index (elem_offset = chain_offset * D, odd
    offsets included: D = 33, 9 with odd shard starts)."""
    assert_shards_equal_full(device, C, D, parts, fused=False)
    assert_shards_equal_full(device, C, D, parts, fused=False, single_calls=True)


@pytest.mark.parametrize('kind,kw', [('uniform', {}), ('normal', {}), ('normal_zig', {}),
                                     ('gamma', {'shape': 7.5}), ('gamma', {'shape': 0.4})])
def test_stand_alone_fills_are_windows_of_one_global_stream(device, kind, kw):
    """binf_rng_*_f64 with elem_offset: any window [e0, e0 + n) of the stream,
    even or odd start, aligned or not, equals that slice of the whole."""
    N = 70001
    whole = torch.empty(N, dtype=torch.float64, device=device)
    _native.rng_fill(kind, whole, 11, 3, **kw)
    guard = -12345.0
    for e0, n in [(0, 10), (1, 10), (2, 1), (3, 2), (1023, 4100), (2048, 2049), (65537, 4464),
                  (N - 1, 1), (777, 0)]:
        buf = torch.full((n + 4,), guard, dtype=torch.float64, device=device)
        for shift in (1, 2):                 # 8-byte-only and 16-byte aligned windows
            buf.fill_(guard)
            win = buf[shift:shift + n]
            _native.rng_fill(kind, win, 11, 3, elem_offset=e0, **kw)
            assert torch.equal(win, whole[e0:e0 + n]), (kind, e0, n, shift)
            assert float(buf[shift - 1]) == guard and float(buf[shift + n]) == guard
    with pytest.raises(ValueError):
        _native.rng_fill(kind, whole[:4], 11, 3, elem_offset=-1, **kw)


def test_draw_dumps_are_windows_of_the_global_streams(device):
    n, C, D = 2, 40, 200
    p, u = _native.hmc_gauss_rng_draws(n, C, D, 5, 9, device)
    p1, u1 = _native.hmc_gauss_rng_draws(n, 7, D, 5, 9, device, chain_offset=13)
    assert torch.equal(p1, p[:, 13:20]) and torch.equal(u1, u[:, 13:20])
    P, U = _native.hmc_gauss_big_rng_draws(6, 8200, 5, 9, device)
    P1, U1 = _native.hmc_gauss_big_rng_draws(2, 8200, 5, 9, device, chain_offset=3)
    assert torch.equal(P1, P[3:5]) and torch.equal(U1, U[3:5])
    with pytest.raises(ValueError):
        _native.hmc_gauss_rng_draws(n, 7, D, 5, 9, device, chain_offset=-1)


def test_same_seed_same_offset_duplicates_chains_and_the_docs_say_so(device):
    """The footgun the offsets remove: two ranks with one seed and NO offset draw
    identical chains."""
    q0 = torch.zeros((2048, 64), dtype=torch.float64, device=device)
    a = run(q0, DeviceRNG(3, device), 2)
    b = run(q0, DeviceRNG(3, device), 2)
    c = run(q0, DeviceRNG(3, device, chain_offset=2048), 2)
    assert torch.equal(a[0], b[0]) and not torch.equal(a[0], c[0])
    assert 'chain_offset' in DeviceRNG.__doc__

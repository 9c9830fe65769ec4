"""GPU tests of the device random-draw kernels (throughput mode; replaces
np.random.normal / uniform / gamma of hmc.py:146,151 and samplers.py:47 when
numpy-stream parity is not required)."""
import numpy as np
import pytest
import torch

from binf_amd import _native
from binf_amd.pdf import native_gauss
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG

pytestmark = pytest.mark.gpu


def host_uniforms(n, seed, offset):
    """Independent restatement: counter (i, offset), key = seed, two 53-bit
    uniforms per block."""
    out = np.empty(2 * ((n + 1) // 2))
    for i in range((n + 1) // 2):
        r = _native.philox4x32_10([i & 0xffffffff, i >> 32, offset & 0xffffffff, offset >> 32],
                                  [seed & 0xffffffff, seed >> 32])
        out[2 * i] = ((r[0] >> 5) * 67108864.0 + (r[1] >> 6)) / 9007199254740992.0
        out[2 * i + 1] = ((r[2] >> 5) * 67108864.0 + (r[3] >> 6)) / 9007199254740992.0
    return out[:n]


def fill(kind, n, seed, offset, device, **kw):
    out = torch.empty(n, dtype=torch.float64, device=device)
    _native.rng_fill(kind, out, seed, offset, **kw)
    return out.cpu().numpy()


def test_uniform_bits_match_host_philox(device):
    for n, seed, off in [(1, 0, 0), (7, 12345, 3), (1000, 2 ** 40 + 17, 2 ** 33 + 5)]:
        assert np.array_equal(fill('uniform', n, seed, off, device), host_uniforms(n, seed, off))


def test_draws_do_not_depend_on_the_launch_size(device):
    for kind, kw in (('uniform', {}), ('normal', {}), ('gamma', {'shape': 10.0})):
        a = fill(kind, 100000, 5, 9, device, **kw)
        b = fill(kind, 10, 5, 9, device, **kw)
        assert np.array_equal(a[:10], b)
        assert not np.array_equal(a[:10], fill(kind, 10, 5, 10, device, **kw))
        assert not np.array_equal(a[:10], fill(kind, 10, 6, 9, device, **kw))


def test_normal_is_box_muller_of_the_uniforms(device):
    n = 2000
    u = host_uniforms(n, 77, 1)
    r = np.sqrt(-2.0 * np.log(1.0 - u[0::2]))
    want = np.empty(n)
    want[0::2] = r * np.cos(2 * np.pi * u[1::2])
    want[1::2] = r * np.sin(2 * np.pi * u[1::2])
    got = fill('normal', n, 77, 1, device)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-13)


def test_moments(device):
    n = 4_000_000
    u = fill('uniform', n, 1, 0, device)
    assert 0.0 <= u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 5e-4 and abs(u.var() - 1 / 12) < 5e-4
    z = fill('normal', n, 1, 1, device)
    assert abs(z.mean()) < 3e-3 and abs(z.var() - 1.0) < 5e-3
    assert abs((z ** 3).mean()) < 1e-2 and abs((z ** 4).mean() - 3.0) < 3e-2
    assert abs(np.corrcoef(z[0::2], z[1::2])[0, 1]) < 3e-3
    for shape in (0.5, 1.0, 10.0, 8192.0):
        g = fill('gamma', 1_000_000, 2, 7, device, shape=shape)
        assert g.min() > 0
        assert abs(g.mean() - shape) < 5 * np.sqrt(shape / 1e6) + 1e-3 * shape
        assert abs(g.var() - shape) < 0.02 * shape + 0.01


def test_device_rng_object_advances_and_feeds_the_sampler(device):
    rng = DeviceRNG(3, device)
    a = rng.normal((4, 8), device)
    b = rng.normal((4, 8), device)
    assert not torch.equal(a, b) and rng.offset == 2
    g = rng.gamma(10.0, 16, device)
    assert g.shape == (16,) and rng.offset == 130
    again = DeviceRNG(3, device)
    assert torch.equal(again.normal((4, 8), device), a)
    # a stationary Gaussian sampled with device draws keeps unit variance
    C, D = 512, 256
    s = HMCSampler(IsotropicGaussian(), torch.zeros((C, D), dtype=torch.float64, device=device),
                   0.2, 10, variable_name='x', rng=DeviceRNG(11, device))
    s.sample_n(20, record=False)
    rec = s.sample_n(40, thin=4)
    var = float(rec.var())
    assert 0.9 < var < 1.1
    assert 0.5 < float(s.acceptance_rate.mean()) <= 1.0


def test_ziggurat_normals(device):
    """1024-layer ziggurat: distribution (moments, KS distance, tails, layer
    boundaries), determinism and launch-size independence."""
    from scipy import stats
    n = 8_000_000
    z = fill('normal_zig', n, 21, 5, device)
    assert np.array_equal(z[:1000], fill('normal_zig', 1000, 21, 5, device))
    assert not np.array_equal(z[:1000], fill('normal_zig', 1000, 21, 6, device))
    assert abs(z.mean()) < 2e-3 and abs(z.var() - 1.0) < 3e-3
    assert abs((z ** 3).mean()) < 6e-3 and abs((z ** 4).mean() - 3.0) < 2e-2
    assert abs((z ** 6).mean() - 15.0) < 0.3
    # Kolmogorov-Smirnov distance on a subsample (critical value ~ 1.63/sqrt(n) at 1 %)
    sub = z[:1_000_000]
    d = stats.kstest(sub, 'norm').statistic
    assert d < 1.63 / np.sqrt(sub.size)
    # tails, including the region beyond the base layer edge R = 4.039
    for t in (1.0, 2.0, 3.0, 4.038849846109505, 4.5):
        want = 2 * stats.norm.sf(t)
        got = (np.abs(z) > t).mean()
        assert abs(got - want) < 5 * np.sqrt(want / n) + 1e-7, (t, got, want)
    assert np.abs(z).max() < 7.0
    assert abs(np.corrcoef(z[0::2], z[1::2])[0, 1]) < 2e-3
    assert abs((z > 0).mean() - 0.5) < 1e-3
    rng = DeviceRNG(3, device)                       # ziggurat is the default
    a = rng.normal((8, 16), device)
    b = DeviceRNG(3, device, normal='box_muller').normal((8, 16), device)
    assert a.shape == b.shape and not torch.equal(a, b)


# ---------------------------------------------------------------------------
# the generator fused into the sampling kernel (csrc/xoshiro.hpp,
# csrc/hmc_gauss_rng.hip): hmc.py:146,151 inside the launch
# ---------------------------------------------------------------------------
def _zig_table():
    import os
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        'binf_amd', 'csrc', 'zig_tables.hpp')
    text = open(path).read()
    body = re.search(r'ZIG_X\[1025\] = \{(.*?)\};', text, re.S).group(1)
    return np.array([float.fromhex(t) for t in body.replace(',', ' ').split()])


class _Xo128(object):
    """xoshiro128++ (Blackman & Vigna), restated on the host."""
    M = 0xffffffff

    def __init__(self, stream, seed, offset):
        r = _native.philox4x32_10([stream & self.M, stream >> 32, offset & self.M, offset >> 32],
                                  [seed & self.M, (seed >> 32) ^ 0x58534f52])
        self.s = [int(v) for v in r]
        if not any(self.s):
            self.s[0] = 1

    @staticmethod
    def _rotl(x, k):
        return ((x << k) | (x >> (32 - k))) & 0xffffffff

    def next(self):
        s = self.s
        r = (self._rotl((s[0] + s[3]) & self.M, 7) + s[0]) & self.M
        t = (s[1] << 9) & self.M
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = self._rotl(s[3], 11)
        return r

    def uniform53(self):
        a, b = self.next(), self.next()
        return ((a >> 5) * 67108864.0 + (b >> 6)) / 9007199254740992.0


def test_fused_generator_bits_match_a_host_restatement(device):
    """D = 8: every lane owns ONE element, so its stream is: one ziggurat
    candidate (two outputs), its resolution if it fails, then the uniform.  The
    host restatement covers the fast path (99.6 % of the lanes) bit for bit."""
    zx = _zig_table()
    assert zx.shape == (1025,) and zx[1024] == 0.0 and zx[1] == 4.038849846109505
    C, D, seed, offset = 300, 8, 2 ** 40 + 12345, 7
    p0, u = _native.hmc_gauss_rng_draws(1, C, D, seed, offset, device)
    p0, u = p0.cpu().numpy()[0], u.cpu().numpy()[0]
    checked = 0
    for c in range(C):
        for j in range(D):
            g = _Xo128(c * 8 + j, seed, offset)           # stream = chain * 8 lanes + accumulator
            hi, lo = g.next(), g.next()
            layer = hi >> 22
            bits = ((0x3ff00000 | (hi & 0xfffff)) << 32) | lo
            d = np.frombuffer(np.uint64(bits).tobytes(), dtype=np.float64)[0]
            x = (2.0 * d - 3.0) * zx[layer]
            if abs(x) < zx[layer + 1]:
                assert p0[c, j] == x, (c, j)
                checked += 1
                if j == 0:
                    assert u[c] == g.uniform53(), c
    assert checked > 0.99 * C * D


FUSED_SHAPES = [(64, 1024, 20, 1.0, 0.0, 'exact'), (70, 768, 5, 2.5, 0.3, 'exact'),
                (5, 33, 7, 1.0, 0.0, 'fma'), (9, 200, 3, 1.0, -0.2, 'exact'),
                (4, 1000, 2, 1.0, 0.0, 'exact'), (3, 920, 2, 2.5, 0.3, 'fma'),
                (130, 7, 4, 1.0, 0.0, 'exact'), (17, 1, 3, 1.0, 0.0, 'exact'),
                (6, 258, 2, 1.0, 0.0, 'exact'), (2100, 1024, 2, 1.0, 0.0, 'exact'),
                # chains of 2 / 4 / 8 waves (the acceptance draw crosses waves)
                (5, 2048, 3, 1.0, 0.0, 'exact'), (3, 1023, 2, 2.5, 0.3, 'exact'),
                (7, 3000, 2, 1.0, 0.0, 'fma'), (3, 4096, 2, 1.0, 0.0, 'exact'),
                (9, 8192, 2, 1.0, 0.0, 'exact'), (2, 7000, 2, 2.5, -0.2, 'exact')]


@pytest.mark.parametrize('C,D,L,k,x0,mode', FUSED_SHAPES)
def test_fused_generator_equals_sampling_from_its_own_dump(device, C, D, L, k, x0, mode):
    """sample_n with the draws generated in the kernel == sample_n fed with the
    dump of the same stream: states, flags, energies, adapted step sizes, all
    bit for bit (regular and ragged trees, several chains per wave, adaption)."""
    n, seed = 5, 99
    q0 = torch.from_numpy(np.random.RandomState(D).standard_normal((C, D))).to(device)
    dt = 0.9 / np.sqrt(k * max(D, 4))
    kw = dict(timestep_adaption_limit=4, variable_name='x', mode=mode, record_energies=True)
    a = HMCSampler(IsotropicGaussian(k, x0), q0, dt, L, rng=DeviceRNG(seed, device, fused='always'), **kw)
    assert native_gauss.fused_rng(a, 'x', D) and native_gauss.draws_in_kernel(a, C, D)
    rec_a = a.sample_n(n)                                    # stream positions 0 .. n - 1
    rec_a2 = a.sample_n(2, thin=2)                           # positions n, n + 1 (one per TRANSITION, ABI 5)
    assert a.rng.offset == n + 2
    p0, u = _native.hmc_gauss_rng_draws(n, C, D, seed, 0, device)
    p1, u1 = _native.hmc_gauss_rng_draws(2, C, D, seed, n, device)
    b = HMCSampler(IsotropicGaussian(k, x0), q0, dt, L, **kw)
    rec_b = b.sample_n(n, p0=p0, u=u)
    e_b = (b.last_e_before.clone(), b.last_e_after.clone())
    h_b = b.accepted_history.clone()
    rec_b2 = b.sample_n(2, thin=2, p0=p1, u=u1)
    assert torch.equal(rec_a, rec_b) and torch.equal(rec_a2, rec_b2)
    assert torch.equal(a.state, b.state) and torch.equal(a.n_accepted, b.n_accepted)
    assert torch.equal(a.timestep, b.timestep)
    assert torch.equal(a.last_e_after, b.last_e_after)
    # single sample() calls take the same route
    a2 = HMCSampler(IsotropicGaussian(k, x0), q0, dt, L, rng=DeviceRNG(seed, device, fused='always'), **kw)
    x = a2.sample()
    assert torch.equal(x, rec_b[0]) and torch.equal(a2.last_move_accepted, h_b[0])
    assert torch.equal(a2.last_e_before, e_b[0][0])
    assert 0 < float(h_b.double().mean()) <= 1.0


def test_fused_generator_stream_properties(device):
    """The dump of the in-kernel stream at C2's shape: determinism, independence
    of the batch size, distribution (moments, KS, tails beyond the base layer),
    no correlation between the elements a lane draws in sequence, between
    neighbouring lanes, between consecutive transitions."""
    from scipy import stats
    C, D, n = 4096, 1024, 2
    p0, u = _native.hmc_gauss_rng_draws(n, C, D, 21, 5, device)
    again, _ = _native.hmc_gauss_rng_draws(n, C, D, 21, 5, device)
    assert torch.equal(p0, again)
    small, us = _native.hmc_gauss_rng_draws(n, 10, D, 21, 5, device)
    assert torch.equal(small, p0[:, :10]) and torch.equal(us, u[:, :10])
    other, uo = _native.hmc_gauss_rng_draws(n, 10, D, 21, 6, device)
    assert not torch.equal(other, small) and not torch.equal(uo, us)
    other, _ = _native.hmc_gauss_rng_draws(n, 10, D, 22, 5, device)
    assert not torch.equal(other, small)
    z = p0.cpu().numpy()
    uu = u.cpu().numpy().reshape(-1)
    N = z.size
    flat = z.reshape(-1)
    assert abs(flat.mean()) < 2e-3 and abs(flat.var() - 1.0) < 3e-3
    assert abs((flat ** 3).mean()) < 6e-3 and abs((flat ** 4).mean() - 3.0) < 2e-2
    assert abs((flat ** 6).mean() - 15.0) < 0.3
    sub = flat[::8][:1_000_000]
    assert stats.kstest(sub, 'norm').statistic < 1.63 / np.sqrt(sub.size)
    for t in (1.0, 2.0, 3.0, 4.038849846109505, 4.5):
        want = 2 * stats.norm.sf(t)
        got = (np.abs(flat) > t).mean()
        assert abs(got - want) < 5 * np.sqrt(want / N) + 1e-7, (t, got, want)
    assert np.abs(flat).max() < 7.0 and abs((flat > 0).mean() - 0.5) < 1e-3
    lim = 5 / np.sqrt(N / 2)
    assert abs(np.corrcoef(z[0, :, :-1].ravel(), z[0, :, 1:].ravel())[0, 1]) < lim   # neighbour lanes
    assert abs(np.corrcoef(z[0, :, :-8].ravel(), z[0, :, 8:].ravel())[0, 1]) < lim   # same lane, next draw
    assert abs(np.corrcoef(z[0].ravel(), z[1].ravel())[0, 1]) < lim                  # next transition
    assert abs(np.corrcoef(z[0, :-1].ravel(), z[0, 1:].ravel())[0, 1]) < lim         # next chain
    assert 0.0 <= uu.min() and uu.max() < 1.0
    assert abs(uu.mean() - 0.5) < 5 / np.sqrt(12 * uu.size)
    assert stats.kstest(uu, 'uniform').statistic < 1.63 / np.sqrt(uu.size)


def test_fused_generator_limits_and_fallback(device):
    z = torch.zeros((3, 9000), dtype=torch.float64, device=device)
    with pytest.raises(NotImplementedError):
        _native.hmc_gauss_rng_draws(1, 3, 9000, 0, 0, device)      # the persistent kernel's entry
    # longer chains draw inside the chunked kernels (test below)
    s = HMCSampler(IsotropicGaussian(), z, 0.02, 3, variable_name='x', rng=DeviceRNG(1, device))
    assert native_gauss.fused_rng(s, 'x', 9000)
    assert s.sample_n(2).shape == (2, 3, 9000) and s.rng.offset == 2
    # few chains of D = 1024: a chain is spread over 4 waves with draws from HBM, which beats
    # the one-wave kernel with its own generator -> the SAME lane-stream draws, written out
    # by the draw kernel first (the choice of launch never changes what a chain draws)
    for C, want in ((64, False), (1024, False), (1025, True), (2100, True), (5000, True)):
        s = HMCSampler(IsotropicGaussian(), torch.zeros((C, 1024), dtype=torch.float64, device=device),
                       0.02, 3, variable_name='x', rng=DeviceRNG(1, device))
        assert native_gauss.fused_rng(s, 'x', 1024) and native_gauss.draws_in_kernel(s, C, 1024) is want, C
    assert _native.gauss_waves_per_chain(512, 1024) == 4
    assert _native.gauss_waves_per_chain(2048, 1024) == 2
    assert _native.gauss_waves_per_chain(4096, 1024) == 1
    assert _native.gauss_waves_per_chain(10, 4096) == 4 and _native.gauss_waves_per_chain(10, 9000) == 0
    # ... and so does a generator that was told not to fuse
    s = HMCSampler(IsotropicGaussian(), z[:, :64].contiguous(), 0.02, 3, variable_name='x',
                   rng=DeviceRNG(1, device, fused=False))
    assert not native_gauss.fused_rng(s, 'x', 64)
    assert s.sample().shape == (3, 64)


@pytest.mark.parametrize('C,D,L,k,x0,mode,adapt', [
    (3, 8193, 2, 1.0, 0.0, 'exact', False), (5, 20000, 3, 2.5, 0.3, 'exact', True),
    (2, 16384, 2, 1.0, 0.0, 'fma', False), (4, 7689, 2, 1.0, 0.0, 'exact', True),
    (70, 8192 * 2 + 5, 1, 1.0, -0.1, 'exact', False)])
def test_long_chain_generator_equals_sampling_from_its_own_dump(device, C, D, L, k, x0, mode, adapt):
    """Chains beyond the persistent kernel (csrc/hmc_gauss_big.hip) with the draws
    generated in the kernels == the same kernels fed with the dump of that stream,
    bit for bit, over consecutive calls (one stream position per call)."""
    seed = 1234
    q0 = torch.from_numpy(np.random.RandomState(D).standard_normal((C, D))).to(device)
    dt = 0.9 / np.sqrt(k * D)
    kw = dict(timestep_adaption_limit=5 if adapt else 0, variable_name='x', mode=mode,
              record_energies=True)
    a = HMCSampler(IsotropicGaussian(k, x0), q0, dt, L, rng=DeviceRNG(seed, device), **kw)
    b = HMCSampler(IsotropicGaussian(k, x0), q0, dt, L, **kw)
    assert native_gauss.fused_rng(a, 'x', D) and not _native.gauss_persist_covers(D)
    for call in range(3):
        xa = a.sample()
        p0, u = _native.hmc_gauss_big_rng_draws(C, D, seed, call, device)
        xb = b.sample(p0=p0, u=u)
        assert torch.equal(xa, xb), call
        assert torch.equal(a.last_move_accepted, b.last_move_accepted)
        assert torch.equal(a.last_e_before, b.last_e_before)
        assert torch.equal(a.last_e_after, b.last_e_after)
        if adapt:
            assert torch.equal(a.timestep, b.timestep)
    assert a.rng.offset == 3 and torch.equal(a.n_accepted, b.n_accepted)
    # the dump: every element written, plausible, independent of the batch size
    p0, u = _native.hmc_gauss_big_rng_draws(C, D, seed, 7, device)
    p1, u1 = _native.hmc_gauss_big_rng_draws(min(C, 2), D, seed, 7, device)
    assert torch.equal(p0[:min(C, 2)], p1) and torch.equal(u[:min(C, 2)], u1)
    z = p0.cpu().numpy()
    assert np.isfinite(z).all() and np.abs(z).max() < 7.5
    if z.size > 50000:
        assert abs(z.mean()) < 5 / np.sqrt(z.size) and abs(z.var() - 1.0) < 8 / np.sqrt(z.size)
    uu = u.cpu().numpy()
    assert (0.0 <= uu).all() and (uu < 1.0).all()


@pytest.mark.parametrize('C,D,coff', [(1, 1, 0), (3, 7, 0), (5, 768, 0), (256, 768, 0), (4096, 1024, 0), (7, 33, 11)])
def test_the_two_draws_of_a_transition_in_one_launch(device, C, D, coff):
    """DeviceRNG.normal_uniform / binf_rng_normal_zig_uniform_f64: the values and stream
    positions of normal() followed by uniform(), for shards too; a Box-Muller generator
    falls back to the two calls; the C entry point refuses equal offsets."""
    a, b = DeviceRNG(5, device, chain_offset=coff), DeviceRNG(5, device, chain_offset=coff)
    a.offset = b.offset = 9
    for _ in range(2):
        p1, u1 = a.normal((C, D), device), a.uniform(C, device)
        p2, u2 = b.normal_uniform((C, D), C, device)
        assert torch.equal(p1, p2) and torch.equal(u1, u2) and a.offset == b.offset
    bm1, bm2 = DeviceRNG(5, device, normal='box_muller'), DeviceRNG(5, device, normal='box_muller')
    p1, u1 = bm1.normal((C, D), device), bm1.uniform(C, device)
    p2, u2 = bm2.normal_uniform((C, D), C, device)
    assert torch.equal(p1, p2) and torch.equal(u1, u2) and bm1.offset == bm2.offset
    buf = torch.empty(4, dtype=torch.float64, device=device)
    rc = _native.lib().binf_rng_normal_zig_uniform_f64(buf.data_ptr(), 4, buf.data_ptr(), 4, 1, 3, 3, 0, 0,
                                                       _native.stream_handle(device))
    assert rc == _native.E_ARG

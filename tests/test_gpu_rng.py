"""GPU tests of the device random-draw kernels (throughput mode; replaces
np.random.normal / uniform / gamma of hmc.py:146,151 and samplers.py:47 when
numpy-stream parity is not required)."""
import numpy as np
import pytest
import torch

from binf_amd import _native
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

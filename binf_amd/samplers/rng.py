"""
Where the momentum and acceptance draws of ``HMCSampler.sample()`` come from.

The reference consumes the global legacy numpy stream: one
``np.random.normal(size=q.shape)`` then one ``np.random.uniform()`` per
``sample()`` (``binf/samplers/hmc.py:146,151``).
"""
import numpy as np
import torch


class HostLegacyRNG(object):
    """Parity source: draws from the global ``np.random`` stream in the
    reference's order and uploads them.  For one chain this consumes the stream
    exactly as the reference does; for C chains it draws ``normal(size=(C, D))``
    then ``uniform(size=C)``.  Host generation + PCIe copy: not for
    throughput."""

    def normal(self, shape, device):
        return torch.from_numpy(np.random.normal(size=tuple(shape))).to(device)

    def uniform(self, n, device):
        return torch.from_numpy(np.random.uniform(size=int(n))).to(device)

    def fill_normal(self, out):
        out.copy_(torch.from_numpy(np.random.normal(size=tuple(out.shape))))

    def fill_uniform(self, out):
        out.copy_(torch.from_numpy(np.random.uniform(size=out.numel())).reshape(out.shape))


class DeviceRNG(object):
    """Throughput source: draws generated in HBM by the library's Philox
    kernels (``csrc/rng.hip``), so nothing crosses PCIe.  Deterministic in
    (seed, call order), independent of launch geometry; NOT stream-compatible
    with numpy."""

    def __init__(self, seed=0, device='cuda', normal='ziggurat', fused=True):
        if normal not in ('ziggurat', 'box_muller'):
            raise ValueError("normal must be 'ziggurat' or 'box_muller'")
        self._normal_kind = 'normal_zig' if normal == 'ziggurat' else 'normal'
        self.seed = int(seed)
        self.offset = 0
        self.device = torch.device(device)
        # fused: a sampler whose kernel can generate its own draws (HMCSampler
        # on a Gaussian, D <= 8192, more than 1024 chains) asks for a stream
        # position with next_offset() instead of for buffers: the momentum
        # never exists in HBM.  Which generator serves a sampler therefore
        # depends on the batch shape; fused=False = always the stand-alone
        # generator kernels (draws independent of the batch shape);
        # fused='always' = the in-kernel generator whenever the shape is covered.
        self.fused = fused if fused == 'always' else bool(fused)

    def next_offset(self):
        """Reserve one launch worth of the in-kernel generator's stream."""
        o = self.offset
        self.offset += 1
        return o

    def _fill(self, kind, dims, device, advance, **kw):
        from binf_amd import _native
        out = torch.empty(tuple(dims), dtype=torch.float64, device=device)
        _native.rng_fill(kind, out, self.seed, self.offset, **kw)
        self.offset += advance
        return out

    def normal(self, shape, device):
        return self._fill(self._normal_kind, shape, device, 1)

    def fill_normal(self, out):
        """normal() into a caller's contiguous buffer (same stream position rules)."""
        from binf_amd import _native
        _native.rng_fill(self._normal_kind, out, self.seed, self.offset)
        self.offset += 1

    def fill_uniform(self, out):
        from binf_amd import _native
        _native.rng_fill('uniform', out, self.seed, self.offset)
        self.offset += 1

    def uniform(self, n, device):
        return self._fill('uniform', (int(n),), device, 1)

    def gamma(self, shape, n, device):
        """Gamma(shape, 1) variates, one per chain (GammaSampler's ``gamma=``)."""
        return self._fill('gamma', (int(n),), device, 128, shape=shape)

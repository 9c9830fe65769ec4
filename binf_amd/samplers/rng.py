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


class DeviceRNG(object):
    """Throughput source: draws generated in HBM by the device generator, so
    nothing crosses PCIe.  Not stream-compatible with numpy."""

    def __init__(self, seed=0, device='cuda'):
        self._gen = torch.Generator(device=device)
        self._gen.manual_seed(int(seed))

    def normal(self, shape, device):
        return torch.randn(tuple(shape), dtype=torch.float64, device=device,
                           generator=self._gen)

    def uniform(self, n, device):
        return torch.rand(int(n), dtype=torch.float64, device=device,
                          generator=self._gen)

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

    # -- checkpoint (binf_amd/checkpoint.py): the global legacy stream itself ---------------
    def state_dict(self):
        name, key, pos, has_gauss, cached = np.random.get_state()
        return {'generator': name, 'key': torch.from_numpy(key.astype(np.int64)), 'pos': int(pos),
                'has_gauss': int(has_gauss), 'cached_gaussian': float(cached)}

    def load_state_dict(self, d):
        np.random.set_state((d['generator'], np.asarray(d['key'], dtype=np.uint32), int(d['pos']),
                             int(d['has_gauss']), float(d['cached_gaussian'])))


class DeviceRNG(object):
    """Throughput source: draws generated on the device, so nothing crosses
    PCIe.  NOT stream-compatible with numpy.

    What a chain draws is a function of ``(seed, call order, GLOBAL chain
    index)`` only -- not of the batch size, the launch geometry, or how a run is
    sharded over GPUs.  ``chain_offset`` is the global index of the first chain
    this process owns (0 for an unsharded run; :meth:`for_shard` derives it from
    ``binf_amd.dist.shard_chains``): a rank that owns chains ``[s, s + C)``
    of a larger run draws exactly what the one-GPU run draws for those chains,
    so an N-GPU run reproduces the 1-GPU run bit for bit.  Two ranks with the
    same seed and the same ``chain_offset`` WOULD produce duplicate chains --
    give every rank its shard's offset (or a different seed).

    Two generators serve the draws:

    * the stand-alone Philox4x32-10 kernels (``csrc/rng.hip``): element ``i`` of a
      ``[C_total x D]`` normal draw is a function of its global flat index;
    * the lane streams of the fused Gaussian HMC kernels (``csrc/xoshiro.hpp``,
      ``fused=True``, the default): a sampler whose kernel can generate its own
      draws (``HMCSampler`` on a Gaussian) asks for a stream position with
      :meth:`next_offset` instead of for buffers.  Large batches generate the
      draws inside the sampling kernel (the momentum never exists in HBM);
      small batches, which run faster with a chain spread over several waves
      (``csrc/hmc_gauss_split.hip``), get THE SAME draws written out by
      ``binf_hmc_gauss_rng_draws_f64`` first.  Either way a seed identifies the
      draws whatever the batch size.

    ``fused=False`` uses the Philox kernels for everything; ``fused='always'``
    generates in the sampling kernel whenever the shape is covered.  Every
    transition takes one stream position, so ``sample_n(n)`` draws exactly what n
    ``sample()`` calls draw (persistent kernel and long-chain path alike).
    """

    def __init__(self, seed=0, device='cuda', normal='ziggurat', fused=True, chain_offset=0):
        if normal not in ('ziggurat', 'box_muller'):
            raise ValueError("normal must be 'ziggurat' or 'box_muller'")
        if int(chain_offset) < 0:
            raise ValueError('chain_offset must be >= 0')
        self._normal_kind = 'normal_zig' if normal == 'ziggurat' else 'normal'
        self.seed = int(seed)
        self.offset = 0
        self.chain_offset = int(chain_offset)
        self.device = torch.device(device)
        self.fused = fused if fused == 'always' else bool(fused)

    @classmethod
    def for_shard(cls, seed, n_chains_total, rank=None, world_size=None, **kw):
        """The generator of this rank's shard of ``n_chains_total`` chains
        (``dist.shard_chains``): same seed on every rank, ``chain_offset`` = the
        shard's first chain.  Returns ``(rng, start, count)``."""
        from binf_amd.dist import shard_chains
        start, count = shard_chains(n_chains_total, rank, world_size)
        return cls(seed=seed, chain_offset=start, **kw), start, count

    # -- checkpoint (binf_amd/checkpoint.py): a stream is its seed and its position ----------
    def state_dict(self):
        return {'seed': self.seed, 'offset': int(self.offset), 'chain_offset': self.chain_offset,
                'normal': self._normal_kind, 'fused': self.fused}

    def load_state_dict(self, d):
        if d['normal'] != self._normal_kind or int(d['chain_offset']) != self.chain_offset:
            raise ValueError('DeviceRNG checkpoint is of another stream layout (normal=%s, chain_offset=%s)'
                             % (d['normal'], d['chain_offset']))
        self.seed, self.offset = int(d['seed']), int(d['offset'])

    def next_offset(self):
        """Reserve one launch worth of the in-kernel generator's stream."""
        o = self.offset
        self.offset += 1
        return o

    def _elem_offset(self, shape):
        # dim 0 is the chain axis: the window of this shard in the global array
        per_chain = 1
        for d in tuple(shape)[1:]:
            per_chain *= int(d)
        return self.chain_offset * per_chain

    def _fill(self, kind, dims, device, advance, **kw):
        from binf_amd import _native
        out = torch.empty(tuple(dims), dtype=torch.float64, device=device)
        _native.rng_fill(kind, out, self.seed, self.offset,
                         elem_offset=self._elem_offset(dims), **kw)
        self.offset += advance
        return out

    def normal(self, shape, device):
        return self._fill(self._normal_kind, shape, device, 1)

    def fill_normal(self, out):
        """normal() into a caller's contiguous buffer (same stream position rules)."""
        from binf_amd import _native
        _native.rng_fill(self._normal_kind, out, self.seed, self.offset,
                         elem_offset=self._elem_offset(out.shape))
        self.offset += 1

    def fill_uniform(self, out):
        from binf_amd import _native
        _native.rng_fill('uniform', out, self.seed, self.offset,
                         elem_offset=self._elem_offset(out.shape))
        self.offset += 1

    def uniform(self, n, device):
        return self._fill('uniform', (int(n),), device, 1)

    def normal_uniform(self, shape, n, device):
        """``(normal(shape, device), uniform(n, device))`` -- the two draws of one HMC
        transition, in that order (hmc.py:146,151) -- from ONE launch; the same values,
        the same stream positions as the two calls."""
        if self._normal_kind != 'normal_zig':
            return self.normal(shape, device), self.uniform(n, device)
        from binf_amd import _native
        p = torch.empty(tuple(shape), dtype=torch.float64, device=device)
        u = torch.empty((int(n),), dtype=torch.float64, device=device)
        _native.rng_fill_normal_zig_uniform(p, u, self.seed, self.offset, self.offset + 1,
                                            self._elem_offset(shape), self._elem_offset((int(n),)))
        self.offset += 2
        return p, u

    def fill_normal_uniform(self, p, u):
        """:meth:`normal_uniform` into the caller's contiguous buffers ``p`` ``[C x D]`` and ``u``
        ``[C]`` (the input buffers of a captured HIP graph): same values, same stream positions."""
        if self._normal_kind != 'normal_zig':
            self.fill_normal(p)
            self.fill_uniform(u)
            return
        from binf_amd import _native
        _native.rng_fill_normal_zig_uniform(p, u, self.seed, self.offset, self.offset + 1,
                                            self._elem_offset(p.shape), self._elem_offset(u.shape))
        self.offset += 2

    def gamma(self, shape, n, device):
        """Gamma(shape, 1) variates, one per chain (GammaSampler's ``gamma=``)."""
        return self._fill('gamma', (int(n),), device, 128, shape=shape)

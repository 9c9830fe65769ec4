"""``binf_amd/checkpoint.py`` on the host: the legacy numpy stream, a sample store and the file
format (tensors and plain containers only: read back with ``weights_only=True``).  Samplers are
resumed on the GPU in tests/test_gpu_checkpoint.py."""
import numpy as np
import pytest
import torch

from binf_amd import checkpoint
from binf_amd.dist import SampleStore
from binf_amd.samplers.rng import HostLegacyRNG


def test_host_stream_and_store_round_trip_through_a_file(tmp_path):
    np.random.seed(3)
    np.random.normal(size=7)                       # leaves a cached gaussian behind
    store = SampleStore(6, 4, 3, thin=2, burn_in=1, device='cpu')
    for i in range(6):
        store.record(torch.full((4, 3), float(i), dtype=torch.float64))
    path = str(tmp_path / 'c.pt')
    checkpoint.save(path, stream=HostLegacyRNG(), store=store)
    want = (np.random.normal(size=5), np.random.uniform(size=3))
    np.random.seed(99)
    other = SampleStore(6, 4, 3, thin=2, burn_in=1, device='cpu')
    ckpt = checkpoint.load(path, stream=HostLegacyRNG(), store=other)
    assert set(ckpt) == {'stream', 'store'}
    got = (np.random.normal(size=5), np.random.uniform(size=3))
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert other.n_seen == 6 and other.n_kept == store.n_kept == 3
    assert torch.equal(other.buffer[:3], store.buffer[:3])
    other.record(torch.zeros((4, 3), dtype=torch.float64))          # goes on where the first stopped
    assert other.n_seen == 7 and other.n_kept == 3
    other.record(torch.ones((4, 3), dtype=torch.float64))
    assert other.n_kept == 4
    # nothing but tensors and plain containers in the file
    raw = torch.load(path, weights_only=True)
    assert isinstance(raw['stream']['key'], torch.Tensor) and raw['stream']['generator'] == 'MT19937'


def test_store_checkpoint_must_fit():
    a = SampleStore(2, 4, 3, device='cpu')
    a.record(torch.zeros((4, 3), dtype=torch.float64))
    with pytest.raises(ValueError):
        SampleStore(2, 5, 3, device='cpu').load_state_dict(a.state_dict())

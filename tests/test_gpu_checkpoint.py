"""Checkpoint / resume (``binf_amd/checkpoint.py``): a run saved after some sweeps and resumed
in freshly built objects draws bit for bit what the uninterrupted run draws -- device
generators (streams are seed + position + global chain index), the host legacy stream,
adaption in progress, the sample store."""
import numpy as np
import pytest
import torch

from binf_amd import checkpoint
from binf_amd.dist import SampleStore
from binf_amd.example.likelihood import POLYVAL
from binf_amd.example.misc import make_posterior
from binf_amd.example.samplers import make_hmc_sampler, make_sampler
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers import BinfState
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG, HostLegacyRNG

pytestmark = pytest.mark.gpu


def data():
    rs = np.random.RandomState(5)
    xs = np.linspace(-2, 2, 20)
    return xs, POLYVAL(xs, np.array([2.0, -4.0, 1.0, 1.5])) + 0.6 * rs.standard_normal(20)


@pytest.mark.parametrize('wiring', ['hmc', 'rwmc', 'hmc_per_variable_loop'])
def test_gibbs_run_resumes_bit_for_bit(device, wiring, tmp_path):
    xs, ys = data()
    C = 40

    def build():
        start = BinfState(dict(coefficients=torch.ones((C, 4), dtype=torch.float64, device=device),
                               precision=torch.ones(C, dtype=torch.float64, device=device)))
        post = make_posterior(xs, ys, POLYVAL)
        if wiring == 'rwmc':
            g = make_sampler(post, 0.1, start, rng=DeviceRNG(9, device))
        else:
            g = make_hmc_sampler(post, 0.02, 10, start, rng=DeviceRNG(9, device), timestep_adaption_limit=8)
        if wiring == 'hmc_per_variable_loop':
            g.fused_sweep = False
        return g, SampleStore(20, C, 5, thin=2, burn_in=1, device=device)

    def sweeps(g, store, n):
        for _ in range(n):
            st = g.sample()
            store.record((st.variables['coefficients'], st.variables['precision']))

    a, sa = build()
    sweeps(a, sa, 11)                                      # the uninterrupted run
    b, sb = build()
    sweeps(b, sb, 5)
    path = str(tmp_path / 'run.pt')
    checkpoint.save(path, gibbs=b, store=sb)
    c, sc = build()                                        # fresh objects, e.g. another process
    ckpt = checkpoint.load(path, gibbs=c, store=sc)
    assert sorted(ckpt) == ['gibbs', 'store'] and sc.n_seen == 5
    sweeps(c, sc, 6)
    for name in ('coefficients', 'precision'):
        assert torch.equal(a.state.variables[name], c.state.variables[name]), name
    assert sa.n_kept == sc.n_kept and torch.equal(sa.buffer[:sa.n_kept], sc.buffer[:sc.n_kept])
    ha, hc = a.subsamplers['coefficients'], c.subsamplers['coefficients']
    if wiring == 'rwmc':
        assert torch.equal(ha.acceptance_rate, hc.acceptance_rate)
    else:
        assert torch.equal(ha.n_accepted, hc.n_accepted) and ha.counter == hc.counter == 11
        assert torch.equal(torch.as_tensor(ha.timestep), torch.as_tensor(hc.timestep))   # adaption went on


@pytest.mark.parametrize('D,rng', [(33, 'device'), (1024, 'device'), (9000, 'device'), (33, 'host')])
def test_hmc_sampler_resumes_with_adaption_in_progress(device, D, rng):
    C = 12
    q0 = torch.randn((C, D), dtype=torch.float64, device=device)

    def build():
        r = DeviceRNG(4, device) if rng == 'device' else HostLegacyRNG()
        return HMCSampler(IsotropicGaussian(2.5, 0.3), q0.clone(), 0.2, 5, timestep_adaption_limit=9,
                          variable_name='x', rng=r)

    if rng == 'host':
        np.random.seed(77)
    a = build()
    for _ in range(4):
        a.sample()
    ckpt = checkpoint.state_dict(hmc=a)
    assert all(not t.is_cuda for t in (ckpt['hmc']['state'], ckpt['hmc']['dt_chain']))
    more_a = [a.sample().clone() for _ in range(3)] + [a.sample_n(4)]
    if rng == 'host':
        np.random.seed(123456)                             # the stream is somewhere else meanwhile
    b = build()
    checkpoint.load_state_dict(ckpt, hmc=b)
    more_b = [b.sample().clone() for _ in range(3)] + [b.sample_n(4)]
    for x, y in zip(more_a, more_b):
        assert torch.equal(x, y)
    assert torch.equal(a.n_accepted, b.n_accepted) and a.counter == b.counter == 11
    assert torch.equal(a.timestep, b.timestep)


def test_checkpoint_refuses_what_does_not_fit(device, tmp_path):
    store = SampleStore(4, 3, 2, thin=2, device=device)
    with pytest.raises(ValueError):
        SampleStore(4, 3, 2, thin=3, device=device).load_state_dict(checkpoint.state_dict(s=store)['s'])
    with pytest.raises(ValueError):
        DeviceRNG(1, device, chain_offset=5).load_state_dict(DeviceRNG(1, device).state_dict())
    with pytest.raises(KeyError):
        checkpoint.load_state_dict(checkpoint.state_dict(s=store), other=store)

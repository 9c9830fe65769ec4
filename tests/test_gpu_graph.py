"""``HMCSampler(graph=True)``: the per-step tier's transition (energies, every gradient call,
kicks and drifts, accept) captured once as a HIP graph and replayed -- the prompt's "capture
launch-bound inner loops in hipGraphs" for PDFs that have no fused kernel.  Same bits as the
eager launches, call after call; configurations that change (adaption ending, a parameter
value) get their own capture; a PDF that cannot be captured falls back to eager launches with
a warning."""
import warnings

import numpy as np
import pytest
import torch

from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers import hmc as H
from binf_amd.samplers.hmc import HMCSampler
from oracle import c_oracle

pytestmark = pytest.mark.gpu


def dev_t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)


class DoubleWell(object):
    """a user's torch PDF (examples/custom_pdf.py)"""

    def __init__(self, a=2.0):
        self.a = a

    def log_prob(self, x):
        w = x * x - 1.0
        return (-self.a) * (w * w).sum(dim=1)

    def gradient(self, x):
        return (4.0 * self.a) * x * (x * x - 1.0)


def plain_gaussian(k=1.0, x0=0.0):
    pdf = IsotropicGaussian(k, x0)
    pdf.native_hmc_spec = lambda name: None          # the per-step tier, library kernels inside
    return pdf


@pytest.mark.parametrize('limit', [0, 4])
def test_graph_replay_equals_eager_launches_call_after_call(device, limit):
    rs = np.random.RandomState(1)
    C, D, L, n = 37, 96, 6, 7
    q0 = rs.standard_normal((C, D)) * 0.4 + 1.0
    p0, u = rs.standard_normal((n, C, D)), rs.uniform(size=(n, C))

    def run(graph):
        s = HMCSampler(DoubleWell(), dev_t(q0, device), 0.12, L, variable_name='x', record_energies=True,
                       timestep_adaption_limit=limit, graph=graph)
        rows = []
        for i in range(n):
            x = s.sample(p0=dev_t(p0[i], device), u=dev_t(u[i], device))
            rows.append((x, s.last_move_accepted.clone(), s.last_e_before, s.last_e_after))
        return s, rows

    se, eager = run(False)
    sg, graph = run(True)
    # adaption on for calls 1 .. limit-1, off afterwards: two configurations, each warmed eagerly
    # once and captured at its second use
    assert len(sg._graphs) == (2 if limit else 1) and not se._graphs
    for (a, fa, ba, aa), (b, fb, bb, ab) in zip(eager, graph):
        assert torch.equal(a, b) and torch.equal(fa, fb) and torch.equal(ba, bb) and torch.equal(aa, ab)
    assert torch.equal(se.n_accepted, sg.n_accepted) and se.counter == sg.counter == n
    assert 0 < int(sg.n_accepted.sum()) < n * C               # rejections happen
    if limit:
        assert torch.equal(se.timestep, sg.timestep)
    # a state handed out is never written again: the rows kept above still differ call to call
    assert not torch.equal(graph[-1][0], graph[-2][0])
    assert graph[-1][0].data_ptr() != graph[-2][0].data_ptr()


def test_graph_mode_on_library_kernels_vs_oracle_and_a_changed_parameter(device):
    """The Gaussian evaluated as written (row-sum and gradient kernels of the library) under a
    graph: bit-identical to the C oracle; ``pdf['k'].set(...)`` is seen (a new capture), so is a
    new step size."""
    rs = np.random.RandomState(2)
    C, D, L = 20, 300, 4
    q = rs.standard_normal((C, D))
    pdf = plain_gaussian(1.0, 0.0)
    s = HMCSampler(pdf, dev_t(q, device), 0.2, L, variable_name='x', graph=True)
    for i, (k, dt) in enumerate([(1.0, 0.2), (1.0, 0.2), (1.0, 0.2), (2.5, 0.2), (2.5, 0.2), (2.5, 0.2),
                                 (2.5, 0.07), (2.5, 0.07), (2.5, 0.07)]):
        pdf['k'].set(k)
        s.timestep = dt
        p0, u = rs.standard_normal((C, D)), rs.uniform(size=C)
        got = s.sample(p0=dev_t(p0, device), u=dev_t(u, device)).cpu().numpy()
        want = c_oracle.hmc_sample_gauss(q, p0, u, dt, L, k=k, x0=0.0)
        assert np.array_equal(got, want['q_out']), i
        assert np.array_equal(s.last_move_accepted.cpu().numpy(), want['accepted'].astype(bool))
        q = want['q_out']
    assert len(s._graphs) == 3


def test_a_pdf_that_synchronises_falls_back_to_eager_launches(device):
    class Syncing(DoubleWell):
        def gradient(self, x):
            float(x.sum())                            # a host read: illegal while capturing
            return DoubleWell.gradient(self, x)

    rs = np.random.RandomState(3)
    q0, p0, u = rs.standard_normal((8, 16)), rs.standard_normal((3, 8, 16)), rs.uniform(size=(3, 8))
    s = HMCSampler(Syncing(), dev_t(q0, device), 0.1, 3, variable_name='x', graph=True)
    e = HMCSampler(DoubleWell(), dev_t(q0, device), 0.1, 3, variable_name='x')
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        for i in range(3):
            a = s.sample(p0=dev_t(p0[i], device), u=dev_t(u[i], device))
            b = e.sample(p0=dev_t(p0[i], device), u=dev_t(u[i], device))
            assert torch.equal(a, b)
    assert s.graph is False and any('graph mode switched off' in str(x.message) for x in w)
    # the device is fine afterwards
    assert torch.equal(s.sample(p0=dev_t(p0[0], device), u=dev_t(u[0], device)),
                       e.sample(p0=dev_t(p0[0], device), u=dev_t(u[0], device)))


def test_parameters_replaced_every_call_switch_graph_mode_off(device):
    """A loop that REPLACES a parameter tensor each call (what GibbsSampler does with the other
    variables' values) never reuses a configuration: graph mode gives up, results unaffected."""
    rs = np.random.RandomState(4)
    q = rs.standard_normal((6, 10))
    pdf = plain_gaussian(1.0, 0.0)
    s = HMCSampler(pdf, dev_t(q, device), 0.1, 2, variable_name='x', graph=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        for i in range(4 * H.GRAPH_MAX_CAPTURES + 3):
            k = 1.0 + 0.01 * i
            pdf['k'].set(k)
            p0, u = rs.standard_normal((6, 10)), rs.uniform(size=6)
            got = s.sample(p0=dev_t(p0, device), u=dev_t(u, device)).cpu().numpy()
            q = c_oracle.hmc_sample_gauss(q, p0, u, 0.1, 2, k=k)['q_out']
            assert np.array_equal(got, q)
    assert s.graph is False and any('none twice' in str(x.message) for x in w)


def test_big_batches_stay_eager_and_the_flag_is_validated(device):
    C, D = 64, H.GRAPH_MAX_ELEMENTS // 64 + 64
    s = HMCSampler(DoubleWell(), torch.ones((C, D), dtype=torch.float64, device=device), 0.01, 2,
                   variable_name='x', graph=True)
    for _ in range(3):
        s.sample()
    assert not s._graphs and not s._graph_warm
    with pytest.raises(ValueError):
        HMCSampler(DoubleWell(), torch.ones((2, 2), dtype=torch.float64, device=device), 0.01, 2, graph='yes')
    # a PDF with a whole-transition kernel takes that kernel, graph or not
    f = HMCSampler(IsotropicGaussian(), torch.ones((4, 8), dtype=torch.float64, device=device), 0.1, 2,
                   variable_name='x', graph='always')
    for _ in range(3):
        f.sample()
    assert not f._graphs


def test_gibbs_keeps_conditional_parameters_in_place_for_a_graphed_subsampler(device):
    """Gibbs-within-HMC on the per-step tier (the polynomial model with its whole-transition
    kernel switched off): with ``graph=True`` on the HMC subsampler the GibbsSampler refreshes the
    conditionals' tensors in place, the transition is captured once and replayed every sweep --
    and every sweep equals the eager run's, bit for bit."""
    from binf_amd.example.likelihood import POLYVAL
    from binf_amd.example.misc import make_posterior
    from binf_amd.example.samplers import make_hmc_sampler
    from binf_amd.samplers import BinfState
    from binf_amd.samplers.rng import DeviceRNG
    rs = np.random.RandomState(5)
    xs = np.linspace(-2, 2, 20)
    ys = POLYVAL(xs, np.array([2.0, -4.0, 1.0, 1.5])) + 0.6 * rs.standard_normal(20)
    C = 48

    def run(graph, sweeps=9):
        start = BinfState(dict(coefficients=torch.ones((C, 4), dtype=torch.float64, device=device),
                               precision=torch.ones(C, dtype=torch.float64, device=device)))
        g = make_hmc_sampler(make_posterior(xs, ys, POLYVAL), 0.02, 12, start, rng=DeviceRNG(3, device),
                             graph=graph)
        g.fused_sweep = False
        hmc = g.subsamplers['coefficients']
        hmc.fused_transition = False                 # the per-step tier: gradient launches + kick / drift
        rows = []
        for _ in range(sweeps):
            st = g.sample()
            rows.append((st.variables['coefficients'].clone(), st.variables['precision'].clone()))
        return g, hmc, rows

    ge, he, eager = run(False)
    gg, hg, graph = run(True)
    assert not he._graphs and len(hg._graphs) == 1 and hg.graph is True
    assert gg._stable() and not ge._stable()
    for (ca, pa), (cb, pb) in zip(eager, graph):
        assert torch.equal(ca, cb) and torch.equal(pa, pb)
    assert torch.equal(he.n_accepted, hg.n_accepted) and 0 < int(hg.n_accepted.sum())
    # the state's tensors are not the conditionals' buffers: nothing handed out is written again
    cond = gg._conditional_pdfs['coefficients']
    ptr = cond['precision'].value.data_ptr()
    assert ptr != gg.state.variables['precision'].data_ptr()
    # switching graph mode on AFTER the Gibbs sampler exists must not turn a state tensor into a
    # buffer either: the first refresh gives the conditional a private copy
    first = ge.state.variables['precision']
    keep = first.clone()
    he.graph = True
    for _ in range(3):
        ge.sample()
    assert torch.equal(first, keep) and ge._stable()
    assert ge._conditional_pdfs['coefficients']['precision'].value.data_ptr() != first.data_ptr()
    gg._update_conditional_pdf_params()                     # what the next sub-step does first
    assert cond['precision'].value.data_ptr() == ptr
    assert torch.equal(cond['precision'].value, gg.state.variables['precision'])


def test_device_generator_draws_straight_into_the_graph_buffers(device):
    """Without supplied draws a graphed sampler lets its DeviceRNG fill the graph's input buffers
    directly (no copy): the same values and stream positions as the eager sampler's draws."""
    from binf_amd.samplers.rng import DeviceRNG
    q0 = torch.randn((50, 40), dtype=torch.float64, device=device) * 0.5
    e = HMCSampler(DoubleWell(), q0.clone(), 0.1, 4, variable_name='x', rng=DeviceRNG(11, device),
                   timestep_adaption_limit=3)
    g = HMCSampler(DoubleWell(), q0.clone(), 0.1, 4, variable_name='x', rng=DeviceRNG(11, device),
                   timestep_adaption_limit=3, graph=True)
    for i in range(8):
        assert torch.equal(e.sample(), g.sample()), i
        assert torch.equal(e.last_move_accepted, g.last_move_accepted)
        assert e.rng.offset == g.rng.offset
    assert len(g._graphs) == 2 and torch.equal(e.n_accepted, g.n_accepted)
    # sample_n on the per-step tier loops over sample(): graphed too
    a, b = e.sample_n(5, thin=2), g.sample_n(5, thin=2)
    assert torch.equal(a, b) and torch.equal(e.accepted_history, g.accepted_history)

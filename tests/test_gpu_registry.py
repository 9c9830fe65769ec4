"""The fused-kernel registry (``binf_amd/native.py``): a model gets a native fast path
without the core knowing it -- the reference's plug-in contract
(``binf/pdf/__init__.py:19-160``, ``binf/model/forwardmodels.py:10-66``) extended to
fused kernels.

A FOURTH kind is registered here, from outside the package: a shifted Gaussian whose
PDF class lives in this test file and whose launcher drives the library's existing
``binf_hmc_sample_gauss_f64``.  ``HMCSampler.sample()`` and ``sample_n()`` take it
with no change to ``HMCSampler`` / ``Posterior`` / ``Likelihood`` / ``GibbsSampler``."""
import numpy as np
import pytest
import torch

from binf_amd import ArrayParameter, _native, native
from binf_amd.params import Parameter
from binf_amd.pdf import AbstractBinfPDF, IsotropicGaussian
from binf_amd.samplers.hmc import _MODES, HMCSampler

pytestmark = pytest.mark.gpu


def dev_t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)


class ShiftedGaussian(AbstractBinfPDF):
    """log p = -0.5 k sum((x - mu)**2) with ``mu`` a scalar: a user's PDF.  Its
    ``_evaluate_*`` run on the library's generic kernels (the per-step tier works without
    any registration); ``native_hmc_spec`` advertises the kind registered below."""

    def __init__(self, k, mu, name='shifted'):
        super(ShiftedGaussian, self).__init__(name=name)
        self._register('k')
        self._register('mu')
        self['k'] = Parameter(k, name='k')
        self['mu'] = Parameter(mu, name='mu')
        self._register_variable('x', differentiable=True)
        self._set_original_variables()
        self.update_var_param_types(x=ArrayParameter)
        self.advertise = True

    def _evaluate_log_prob(self, x):
        x2 = x if x.dim() == 2 else x.reshape(1, -1)
        return _native.row_sum(x2, _native.ROW_SUMSQ_SHIFT, shift=self['mu'].value,
                               scale=-0.5 * self['k'].value)

    def _evaluate_gradient(self, x):
        x2 = x if x.dim() == 2 else x.reshape(1, -1)
        return _native.gauss_grad(x2, self['k'].value, self['mu'].value).view(x.shape)

    def clone(self):
        return self.__class__(self['k'].value, self['mu'].value, self.name)

    def native_hmc_spec(self, variable_name):
        if self.advertise and variable_name == 'x':
            return ('shifted_gauss', float(self['k'].value), float(self['mu'].value))
        return None


CALLS = {'hmc': 0, 'covers': 0}


def _covers(sampler, spec, D, C):
    CALLS['covers'] += 1
    return D <= 4096                       # this launcher: the persistent kernel's range only


def _hmc(sampler, spec, q0, p0, u, accepted, adapt):
    """One transition of every chain through the library's Gaussian kernel, with the
    shift as its x0."""
    CALLS['hmc'] += 1
    _, k, mu = spec
    C, D = q0.shape
    q_out = torch.empty_like(q0)
    eb = torch.empty(C, dtype=torch.float64, device=q0.device)
    ea = torch.empty(C, dtype=torch.float64, device=q0.device)
    _native.hmc_sample_gauss(q0, p0, u, q_out, accepted, sampler.n_accepted, eb, ea,
                             sampler._timestep, sampler._dt_chain, sampler.nsteps, k, mu, adapt,
                             sampler.adaption_uprate, sampler.adaption_downrate, _MODES[sampler.mode])
    sampler.last_e_before, sampler.last_e_after = eb, ea
    return q_out


@pytest.fixture
def fourth_kind():
    CALLS['hmc'] = CALLS['covers'] = 0
    native.register('shifted_gauss', hmc=_hmc, covers=_covers)
    yield native.get('shifted_gauss')
    native.unregister('shifted_gauss')
    assert native.get('shifted_gauss') is None


def test_an_externally_registered_kind_is_taken_by_sample_and_sample_n(device, fourth_kind):
    C, D, L, dt, k, mu = 37, 200, 7, 0.21, 1.7, 0.4
    rs = np.random.RandomState(3)
    q0 = rs.standard_normal((C, D)) + mu
    p0 = rs.standard_normal((4, C, D))
    u = rs.uniform(size=(4, C))

    def run(pdf):
        s = HMCSampler(pdf, dev_t(q0, device), dt, L, variable_name='x', record_energies=True)
        first = s.sample(p0=dev_t(p0[0], device), u=dev_t(u[0], device)).clone()
        e1 = s.last_e_after.clone()
        rec = s.sample_n(3, p0=dev_t(p0[1:], device), u=dev_t(u[1:], device))
        return first, e1, rec, s.accepted_history.clone(), s.n_accepted.clone(), s.counter

    mine = run(ShiftedGaussian(k, mu))
    assert CALLS['hmc'] == 4 and CALLS['covers'] >= 2       # 1 sample() + 3 inside sample_n()
    builtin = run(IsotropicGaussian(k, mu))                  # the library's own kind, same kernel
    for a, b in zip(mine[:5], builtin[:5]):
        assert torch.equal(a, b)
    assert mine[5] == builtin[5] == 4
    # ... and the same PDF on the generic per-step tier (no registration needed for THAT):
    # the same chains, bit for bit -- the kind changes the launch count, not the result
    plain = ShiftedGaussian(k, mu)
    plain.advertise = False
    before = CALLS['hmc']
    generic = run(plain)
    assert CALLS['hmc'] == before
    assert torch.equal(generic[0], mine[0]) and torch.equal(generic[2], mine[2])
    assert torch.equal(generic[3], mine[3])


def test_a_kinds_covers_hook_sends_other_shapes_to_the_per_step_tier(device, fourth_kind):
    C, D = 3, 5000                          # beyond what this kind's covers() accepts
    rs = np.random.RandomState(4)
    s = HMCSampler(ShiftedGaussian(1.0, -0.2), dev_t(rs.standard_normal((C, D)), device), 0.1, 2,
                   variable_name='x')
    assert s._fused_spec('x', D, C) is None and s._fused_spec('x', 100, C) is not None
    s.sample(p0=dev_t(rs.standard_normal((C, D)), device), u=dev_t(rs.uniform(size=C), device))
    assert CALLS['hmc'] == 0 and CALLS['covers'] >= 1


# (the registry's API and the "core never names a model" grep are CPU tests: tests/test_registry.py)


def test_example_kinds_register_on_import_and_dispatch(device):
    """Importing the model classes registers their kinds; the Posterior's spec methods are the
    registry's answers (unchanged tuples: tests/test_gpu_poly.py, test_gpu_distance.py hold the
    kernels to the per-step tier through them)."""
    from binf_amd.example.distance import make_distance_likelihood
    from binf_amd.example.likelihood import POLYVAL, make_likelihood
    from binf_amd.example.priors import GammaPrior, GaussianPrior
    from binf_amd.pdf.posteriors import Posterior
    assert native.get('poly') is not None and native.get('pairdist') is not None
    xs = np.linspace(-1, 1, 20)
    lik = make_likelihood(xs, np.zeros(20), POLYVAL)
    assert lik._native_pair() is not None
    post = Posterior({lik.name: lik}, {'precision_prior': GammaPrior(1.0, 0.2),
                                       'coefficients_prior': GaussianPrior(np.zeros(4), np.ones(4))})
    cond = post.conditional_factory(precision=2.0)
    assert cond.native_hmc_spec('coefficients')[0] == 'poly'
    assert cond.native_leapfrog_spec('coefficients')[0] == 'poly'
    assert cond.native_energy_spec('coefficients') is None
    assert post.native_hmc_spec('coefficients') is None              # precision still free
    n = 8
    dl = make_distance_likelihood(np.ones(n * (n - 1) // 2), n)
    dpost = Posterior({dl.name: dl}, {}).conditional_factory(precision=1.0)
    assert dpost.native_leapfrog_spec('coordinates')[0] == 'pairdist'
    assert dpost.native_energy_spec('coordinates')[0] == 'pairdist'
    assert dpost.native_hmc_spec('coordinates') is None

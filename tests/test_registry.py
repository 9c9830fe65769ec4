"""CPU tests of the fused-kernel registry (``binf_amd/native.py``): the API, that the core never
names a model, and that ``HMCSampler`` / ``Likelihood`` / ``GibbsSampler`` really dispatch through
registered hooks -- with stand-in kinds whose hooks are plain Python on host tensors (no kernel is
launched; the kernels behind the real kinds are tested on the GPU, tests/test_gpu_registry.py)."""
import os
import subprocess

import numpy as np
import pytest
import torch

from binf_amd import ArrayParameter, native
from binf_amd.model.errormodels import AbstractErrorModel
from binf_amd.model.forwardmodels import AbstractForwardModel
from binf_amd.params import Parameter
from binf_amd.pdf import AbstractBinfPDF, IsotropicGaussian
from binf_amd.pdf.likelihoods import Likelihood
from binf_amd.samplers import BinfState
from binf_amd.samplers.gibbs import GibbsSampler
from binf_amd.samplers.hmc import HMCSampler
from conftest import ROOT


def test_registry_api():
    assert 'gauss' in {k.name for k in native.kinds()}
    with pytest.raises(ValueError):
        native.register('gauss', hmc=lambda *a: None)        # taken (replace=True to override)
    with pytest.raises(TypeError):
        native.register('x', no_such_hook=lambda *a: None)
    with pytest.raises(TypeError):
        native.register('x', hmc='not callable')
    with pytest.raises(TypeError):
        native.register('x', likelihood={'polynomial': (None, None)})
    with pytest.raises(TypeError):
        native.register('', hmc=lambda *a: None)
    assert native.get('no such kind') is None and native.get(('gauss', 1.0, 0.0)).name == 'gauss'
    assert native.match('hmc', IsotropicGaussian(), 'x') is None     # a PDF that IS a kind answers itself
    assert IsotropicGaussian(2.0, 0.5).native_hmc_spec('x') == ('gauss', 2.0, 0.5)
    k = native.register('tmp_kind', extras={'answer': 42})
    try:
        assert native.get('tmp_kind') is k and k.extras['answer'] == 42 and 'tmp_kind' in repr(k)
        native.register('tmp_kind', replace=True, hmc=lambda *a: None)
        assert native.get('tmp_kind').hmc is not None
    finally:
        native.unregister('tmp_kind')
    assert native.get('tmp_kind') is None


def test_the_core_never_names_a_model():
    """`grep -rn "binf_amd.example" binf_amd/samplers binf_amd/pdf binf_amd/model`
    is empty -- and so is a search for the kinds' names in the core's code."""
    r = subprocess.run(['grep', '-rn', '--include=*.py', 'binf_amd.example',
                        os.path.join(ROOT, 'binf_amd', 'samplers'), os.path.join(ROOT, 'binf_amd', 'pdf'),
                        os.path.join(ROOT, 'binf_amd', 'model')], stdout=subprocess.PIPE)
    assert r.stdout.decode() == ''
    for rel in ('samplers/hmc.py', 'samplers/gibbs.py', 'pdf/posteriors.py', 'pdf/likelihoods.py'):
        src = open(os.path.join(ROOT, 'binf_amd', rel)).read()
        for name in ("'poly'", "'pairdist'", "'polynomial'", "'gaussian_pairdist'", "'gauss'", 'GAUSS',
                     'IsotropicGaussian(', 'hmc_sample_gauss', 'hmc_sample_n_gauss'):
            assert name not in src, (rel, name)


class _Pdf(AbstractBinfPDF):
    def __init__(self):
        super(_Pdf, self).__init__(name='fake')
        self._register_variable('x', differentiable=True)
        self._set_original_variables()
        self.update_var_param_types(x=ArrayParameter)

    def native_hmc_spec(self, variable_name):
        return ('fake_kind', 3.0) if variable_name == 'x' else None

    def clone(self):
        return self.__class__()


def test_hmc_sampler_dispatches_through_the_registered_hooks():
    calls = []

    def hmc(sampler, spec, q0, p0, u, accepted, adapt):
        calls.append(('hmc', spec, tuple(q0.shape), p0 is not None, u is not None, adapt))
        accepted.fill_(1)
        return q0 + spec[1]

    def covers(sampler, spec, D, C):
        calls.append(('covers', D, C))
        return D <= 8

    q0 = torch.zeros((5, 4), dtype=torch.float64)
    native.register('fake_kind', hmc=hmc, covers=covers)
    try:
        np.random.seed(0)
        s = HMCSampler(_Pdf(), q0, 0.1, 3, variable_name='x')
        assert s._fused_spec('x', 4, 5) == ('fake_kind', 3.0) and s._fused_spec('x', 9, 5) is None
        out = s.sample()
        assert torch.equal(out, q0 + 3.0) and s.counter == 1 and bool(s.last_move_accepted.all())
        assert ('hmc', ('fake_kind', 3.0), (5, 4), True, True, False) in calls
        # no hmc_n hook: sample_n loops over sample(), records every thin-th state
        rec = s.sample_n(4, thin=2)
        assert rec.shape == (2, 5, 4) and torch.equal(rec[1], q0 + 15.0) and s.counter == 5
        assert s.accepted_history.shape == (4, 5)
        # hmc_n and hmc_rng take over when registered
        def hmc_n(sampler, spec, n, thin, p0, u, record, out, q0, shape):
            calls.append(('hmc_n', n, thin, p0 is None, record))
            sampler.counter += n
            return True, (q0 + 100.0, None)

        def hmc_rng(sampler, spec, q0, shape):
            calls.append(('hmc_rng',))
            sampler.counter += 1
            sampler.state = q0 - 1.0
            return sampler.state
        native.register('fake_kind', replace=True, hmc=hmc, covers=covers, hmc_n=hmc_n, hmc_rng=hmc_rng)
        before = s.state.clone()
        assert s.sample_n(7, record=False) is None and torch.equal(s.state, before + 100.0)
        assert ('hmc_n', 7, 1, True, False) in calls
        assert torch.equal(s.sample(), before + 99.0) and calls[-1] == ('hmc_rng',)
        # draws supplied: hmc_rng is not asked
        n_rng = sum(1 for c in calls if c[0] == 'hmc_rng')
        s.sample(p0=torch.zeros_like(q0), u=torch.zeros(5, dtype=torch.float64))
        assert sum(1 for c in calls if c[0] == 'hmc_rng') == n_rng
    finally:
        native.unregister('fake_kind')
    # the kind is gone: the spec is not taken any more
    assert HMCSampler(_Pdf(), q0, 0.1, 3, variable_name='x')._fused_spec('x', 4, 5) is None


class _Fwm(AbstractForwardModel):
    def __init__(self):
        super(_Fwm, self).__init__('fwm')
        self._register_variable('theta', differentiable=True)
        self.update_var_param_types(theta=ArrayParameter)
        self._set_original_variables()

    def _evaluate(self, theta):
        return np.asarray(theta) * 2.0

    def _evaluate_jacobi_matrix(self, theta):
        return 2.0 * np.eye(len(theta))

    def clone(self):
        return self.__class__()

    def native_spec(self):
        return ('fake_fwd', self)


class _Em(AbstractErrorModel):
    def __init__(self):
        super(_Em, self).__init__('em')
        self._register_variable('mock_data')
        self.update_var_param_types(mock_data=ArrayParameter)
        self._set_original_variables()

    def _evaluate_log_prob(self, mock_data):
        return -0.5 * float(np.sum(np.asarray(mock_data) ** 2))

    def _evaluate_gradient(self, mock_data):
        return np.asarray(mock_data)

    def clone(self):
        return self.__class__()

    def native_spec(self):
        return ('fake_err', self)


def test_likelihood_dispatches_to_the_hooks_registered_for_its_model_pair():
    lik = Likelihood('lik', _Fwm(), _Em())
    theta = np.array([1.0, -2.0, 0.5])
    plain_lp, plain_g = lik.log_prob(theta=theta), lik.gradient(theta=theta)
    assert plain_lp == -0.5 * np.sum((2 * theta) ** 2) and np.array_equal(plain_g, 4.0 * theta)
    assert lik._native_pair() is None                     # nobody fuses this pair yet
    seen = []

    def lp(likelihood, fwm, em, fwm_vars, em_vars):
        seen.append(('lp', type(fwm).__name__, type(em).__name__, sorted(fwm_vars), sorted(em_vars)))
        return 123.0

    def grad(likelihood, fwm, em, fwm_vars, em_vars):
        seen.append(('grad',))
        return None                                       # "not this time": the models are evaluated as written
    native.register('fake_pair', likelihood={('fake_fwd', 'fake_err'): (lp, grad)})
    try:
        assert lik._native_pair() is not None
        assert lik.log_prob(theta=theta) == 123.0
        assert np.array_equal(lik.gradient(theta=theta), plain_g)
        assert seen == [('lp', '_Fwm', '_Em', ['theta'], []), ('grad',)]
    finally:
        native.unregister('fake_pair')
    assert lik.log_prob(theta=theta) == plain_lp


def test_gibbs_sampler_offers_its_sweeps_to_the_kinds():
    class Sub(object):
        pdf = None
        state = None

        def __init__(self, step):
            self.step = step

        def sample(self):
            return self.state + self.step

    class Pdf(AbstractBinfPDF):
        def __init__(self):
            super(Pdf, self).__init__(name='p')
            for v in ('a', 'b'):
                self._register_variable(v)
            self._set_original_variables()
            self.update_var_param_types(a=Parameter, b=Parameter)

        def clone(self):
            return self.__class__()

        def conditional_factory(self, **fixed):
            return self.clone()

    g = GibbsSampler(Pdf(), BinfState({'a': 1.0, 'b': 10.0}), {'a': Sub(1.0), 'b': Sub(100.0)})
    assert g.sample().variables == {'a': 2.0, 'b': 110.0}           # the per-variable loop
    offered = []

    def gibbs(gs, n, thin, record):
        offered.append((n, thin, record))
        if n > 1:
            return False, None                                   # only single sweeps, say
        gs._update_state(a=-1.0, b=-2.0)
        return True, None
    native.register('fake_gibbs', gibbs=gibbs)
    try:
        assert g.sample().variables == {'a': -1.0, 'b': -2.0} and offered[-1] == (1, 1, False)
        g.fused_sweep = False                                    # the switch the tests use
        assert g.sample().variables == {'a': 0.0, 'b': 98.0}
        g.fused_sweep = True
        rec = g.sample_n(3, thin=1)                              # refused for n > 1: loops over sample()
        assert (3, 1, True) in offered and rec['a'] == [-1.0, -1.0, -1.0]
    finally:
        native.unregister('fake_gibbs')

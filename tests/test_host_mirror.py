"""CPU tests of the host-side mirror of binf's plug-in surface, pinned by the
known answers the reference's own unit tests hold:

  binf/tests/pdf/__init__.py      -13.0, -29.0, fix / conditional / complete
  binf/tests/pdf/likelihoods.py   252.0, [14ab^2, 22ab^2], parameter binding
  binf/tests/samplers/gibbs.py    sweep x == 3.0, y == 18.0, conditional set-up

The mock classes follow the shape of the reference's mocks so that the tests
read like the reference's."""
import numpy
import pytest

from binf_amd import ArrayParameter, Parameter
from binf_amd.model.errormodels import AbstractErrorModel
from binf_amd.model.forwardmodels import AbstractForwardModel
from binf_amd.params import ParameterNotFoundError
from binf_amd.pdf import AbstractBinfPDF
from binf_amd.pdf.likelihoods import Likelihood
from binf_amd.pdf.posteriors import Posterior
from binf_amd.pdf.priors import AbstractPrior
from binf_amd.samplers import BinfState
from binf_amd.samplers.gibbs import GibbsSampler
from binf_amd.samplers.hmc import HMCSampler


# ---------------------------------------------------------------------------
# parameters
# ---------------------------------------------------------------------------
def test_parameter_binding_propagates_transitively():
    top, mid, leaf = Parameter(1.0, 'p'), Parameter(1.0, 'p'), Parameter(1.0, 'p')
    mid.bind_to(top)
    leaf.bind_to(mid)
    top.set(7.5)
    assert mid.value == 7.5 and leaf.value == 7.5
    other = Parameter(0.0, 'p')
    leaf.bind_to(other)              # re-binding detaches from the old leader
    top.set(1.0)
    assert leaf.value == 7.5 and mid.value == 1.0


def test_array_parameter_wraps_in_numpy_array():
    a = ArrayParameter([1.0, 2.0], 'a')
    assert isinstance(a.value, numpy.ndarray)


# ---------------------------------------------------------------------------
# AbstractBinfPDF  (binf/tests/pdf/__init__.py)
# ---------------------------------------------------------------------------
class MockParameter(Parameter):
    pass


class MockBinfPDF(AbstractBinfPDF):

    def __init__(self, name='MockBinfPDF'):
        super(MockBinfPDF, self).__init__(name=name)
        self._register('ParamA')
        self['ParamA'] = Parameter(2.0, 'ParamA')
        self._register_variable('x')
        self._register_variable('y')
        self.update_var_param_types(x=Parameter, y=Parameter)
        self._set_original_variables()

    def _evaluate_log_prob(self, x, y):
        return -0.5 * self['ParamA'].value * (x ** 2 + y ** 2)

    def clone(self):
        copy = self.__class__()
        copy.set_fixed_variables_from_pdf(self)
        return copy


def test_pdf_fix_variables():
    pdf = MockBinfPDF()
    pdf.fix_variables(y=5.0)
    assert pdf.variables == {'x'}
    assert pdf['y'].value == 5.0
    with pytest.raises(ValueError):
        pdf.fix_variables(z=2.0)


def test_pdf_conditional_factory_known_answers():
    cond = MockBinfPDF().conditional_factory(x=5.0)
    assert 'x' in cond.parameters and cond['x'].value == 5.0
    assert cond.variables == {'y'}
    assert cond.log_prob(y=2.0) == -29.0
    cond2 = cond.conditional_factory(y=2.0)
    assert 'y' in cond2.parameters and cond2['y'].value == 2.0
    assert len(cond2.variables) == 0
    assert cond2.log_prob() == -29.0


def test_pdf_set_fixed_variables_from_pdf():
    pdf1, pdf2 = MockBinfPDF(), MockBinfPDF()
    pdf1.fix_variables(y=2.0)
    pdf2.set_fixed_variables_from_pdf(pdf1)
    assert 'y' in pdf2.parameters and pdf2['y'].value == 2.0


def test_pdf_log_prob_known_answer():
    assert MockBinfPDF().log_prob(x=3, y=2) == -13.0


def test_pdf_update_var_param_types():
    pdf = MockBinfPDF()
    pdf.update_var_param_types(x=MockParameter)
    assert pdf.var_param_types['x'] == MockParameter


def test_pdf_complete_variables():
    pdf = MockBinfPDF()
    pdf.fix_variables(x=7.0)
    variables = {'y': 2.34}
    pdf._complete_variables(variables)
    assert variables == {'y': 2.34, 'x': 7.0}


def test_named_callable_errors():
    pdf = MockBinfPDF()
    with pytest.raises(ValueError):
        pdf._register_variable('x')              # duplicate
    with pytest.raises(ValueError):
        pdf._register_variable(3)                # not a string
    with pytest.raises(ValueError):
        pdf(x=1.0)                               # wrong number of arguments
    with pytest.raises(ParameterNotFoundError):
        pdf['nope']
    pdf2 = MockBinfPDF()
    pdf2._var_param_types.pop('y')
    with pytest.raises(ValueError):              # no parameter type declared
        pdf2.fix_variables(y=1.0)
    with pytest.raises(NotImplementedError):
        pdf.gradient(x=1.0, y=1.0)               # no gradient implemented


# ---------------------------------------------------------------------------
# Likelihood  (binf/tests/pdf/likelihoods.py)
# ---------------------------------------------------------------------------
class MockErrorModel(AbstractErrorModel):

    def __init__(self):
        super(MockErrorModel, self).__init__('StupidErrorModel')
        self._register('ParamB')
        self['ParamB'] = Parameter(4.0, 'ParamB')
        self._register_variable('mock_data', differentiable=True)
        self._register_variable('a')
        self.update_var_param_types(mock_data=ArrayParameter, a=Parameter)
        self._set_original_variables()

    def _evaluate_log_prob(self, mock_data, a):
        return a * numpy.sum(mock_data ** 2)

    def _evaluate_gradient(self, mock_data, a):
        return a * 2.0 * mock_data

    def clone(self):
        copy = self.__class__()
        copy.set_fixed_variables_from_pdf(self)
        return copy


class MockForwardModel(AbstractForwardModel):

    def __init__(self, parameters=()):
        super(MockForwardModel, self).__init__('testfwm', parameters)
        self._register_variable('X')
        self._register_variable('b')
        self.update_var_param_types(X=ArrayParameter, b=Parameter)
        self._set_original_variables()

    def _evaluate(self, X, b):
        return b * numpy.array([1.0, 2.0, 3.0])

    def _evaluate_jacobi_matrix(self, X, b):
        return b * numpy.array([[2.0, 1.0, 1.0],
                                [1.0, 2.0, 2.0]])

    def clone(self):
        pass


class NoAutomaticParamsLikelihood(Likelihood):

    def __init__(self, name, forward_model, error_model):
        AbstractBinfPDF.__init__(self, name)
        self._forward_model = forward_model
        self._error_model = error_model
        self._set_original_variables()


def make_mock_likelihood():
    return Likelihood('testL',
                      MockForwardModel(parameters=[Parameter(2.0, 'ParamA')]),
                      MockErrorModel())


def test_likelihood_setup_parameters_binds_models():
    L = NoAutomaticParamsLikelihood(
        'test', MockForwardModel(parameters=[Parameter(2.0, 'ParamA')]),
        MockErrorModel())
    L._setup_parameters()
    assert 'ParamA' in L.parameters and L['ParamA'].value == 2.0
    L['ParamA'].set(3.0)
    assert L.forward_model['ParamA'].value == 3.0
    assert 'ParamB' in L.parameters and L['ParamB'].value == 4.0
    L['ParamB'].set(7.0)
    assert L.error_model['ParamB'].value == 7.0


def test_likelihood_split_variables():
    fwm, em = make_mock_likelihood()._split_variables(
        {'X': numpy.array([1.0, 2.0]), 'a': 5.0, 'b': 2.0})
    assert set(fwm) == {'X', 'b'} and set(em) == {'a'}


def test_likelihood_log_prob_known_answer():
    L = make_mock_likelihood()
    assert L.log_prob(X=numpy.array([1.2, 4.2, 54.5]), a=2.0, b=3.0) == 252.0


def test_likelihood_gradient_known_answer():
    a, b = 2.0, 3.0
    expected = numpy.array([14 * a * b ** 2, 22 * a * b ** 2])
    got = make_mock_likelihood().gradient(X=numpy.array([1.2, 4.2]), a=a, b=b)
    assert numpy.all(got == expected)


def test_likelihood_variables_exclude_mock_data():
    L = make_mock_likelihood()
    assert L.variables == {'X', 'b', 'a'}
    assert 'mock_data' not in L.variables


# ---------------------------------------------------------------------------
# Posterior (not tested by the reference; checked against its documented rules)
# ---------------------------------------------------------------------------
class FlatPrior(AbstractPrior):
    """log p = c * a, variable 'a' registered NON-differentiable."""

    def __init__(self, c=0.5):
        super(FlatPrior, self).__init__('a_prior')
        self._c = c
        self._register_variable('a')
        self.update_var_param_types(a=Parameter)
        self._set_original_variables()

    def _evaluate_log_prob(self, a):
        return self._c * a

    def _evaluate_gradient(self, a):
        raise AssertionError('must be skipped: no differentiable variable')

    def clone(self):
        copy = self.__class__(self._c)
        copy.set_fixed_variables_from_pdf(self)
        return copy


class DiffForwardModel(MockForwardModel):
    """Same mock, with X registered as differentiable."""

    def __init__(self, parameters=()):
        AbstractForwardModel.__init__(self, 'testfwm', parameters)
        self._register_variable('X', differentiable=True)
        self._register_variable('b')
        self.update_var_param_types(X=ArrayParameter, b=Parameter)
        self._set_original_variables()


def test_posterior_without_differentiable_component_has_no_force():
    L = make_mock_likelihood()           # X registered non-differentiable
    post = Posterior({L.name: L}, {'a_prior': FlatPrior(0.5)})
    # reference posteriors.py:177-187: zeros over the differentiable variables
    # (none here), nothing added
    g = post.gradient(X=numpy.array([1.2, 4.2]), a=2.0, b=3.0)
    assert isinstance(g, numpy.ndarray) and g.shape == (0,)


def test_posterior_sums_components_and_skips_nondifferentiable_ones():
    L = Likelihood('testL', DiffForwardModel(parameters=[Parameter(2.0, 'ParamA')]),
                   MockErrorModel())
    post = Posterior({L.name: L}, {'a_prior': FlatPrior(0.5)})
    assert post.variables == {'X', 'a', 'b'}
    X = numpy.array([1.2, 4.2, 54.5])
    # energy: both components
    assert post.log_prob(X=X, a=2.0, b=3.0) == 0.5 * 2.0 + 252.0
    # force: the prior has no differentiable variable -> skipped (quirk Q4)
    got = post.gradient(X=numpy.array([1.2, 4.2]), a=2.0, b=3.0)
    assert numpy.all(got == numpy.array([252.0, 396.0]))
    # parameters of the components are mirrored and bound
    assert 'ParamA' in post.parameters and 'ParamB' in post.parameters
    post['ParamA'].set(9.0)
    assert L['ParamA'].value == 9.0 and L.forward_model['ParamA'].value == 9.0


# ---------------------------------------------------------------------------
# BinfState
# ---------------------------------------------------------------------------
def test_binfstate_hands_out_copies_of_the_mapping():
    s = BinfState({'x': 1.0})
    v = s.variables
    v['x'] = 99.0
    assert s.variables['x'] == 1.0
    s.update_variables(y=2.0)
    assert s.variables == {'x': 1.0, 'y': 2.0}
    assert s.momenta == {}
    assert BinfState().variables == {}


# ---------------------------------------------------------------------------
# GibbsSampler  (binf/tests/samplers/gibbs.py)
# ---------------------------------------------------------------------------
class MockSampler(object):

    def __init__(self, variable_name):
        self.pdf = None
        self._state = 5.0
        self.variable_name = variable_name

    @property
    def state(self):
        return self._state

    @state.setter
    def state(self, value):
        self._state = value

    @property
    def last_draw_stats(self):
        return {self.variable_name:
                {'testlastdrawstats{}'.format(self.state): self.state}}

    @property
    def sampling_stats(self):
        return {'testsamplingstats{}'.format(self.state): self.state}

    def sample(self):
        if 'y' in self.pdf.parameters:
            return self.state * 2.0 * self.pdf['y'].value
        return self.state * 2.0 * self.pdf['x'].value


def make_gibbs():
    return GibbsSampler(MockBinfPDF(), BinfState({'x': 2.0, 'y': 3.0}),
                        {'x': MockSampler('x'), 'y': MockSampler('y')})


def test_gibbs_setup_conditional_pdfs():
    g = make_gibbs()
    assert set(g._conditional_pdfs) == {'x', 'y'}
    assert g._conditional_pdfs['x']['y'].value == 3.0
    assert g._conditional_pdfs['y']['x'].value == 2.0
    assert g._conditional_pdfs['x'].variables == {'x'}
    assert g.subsamplers['x'].pdf['y'].value == 3.0
    assert g._conditional_pdfs['y'].variables == {'y'}
    assert g.subsamplers['y'].pdf['x'].value == 2.0


def test_gibbs_update_conditional_pdf_params():
    g = make_gibbs()
    g.state.update_variables(x=5.0)
    g._update_conditional_pdf_params()
    assert g._conditional_pdfs['y']['x'].value == 5.0


def test_gibbs_update_samplers():
    g = make_gibbs()
    new = MockSampler('x')
    new.pdf = MockBinfPDF()
    new.pdf['ParamA'].set(23.0)
    g.update_samplers(x=new)
    assert g.subsamplers['x'].pdf['ParamA'].value == 23.0


def test_gibbs_update_subsampler_states_including_an_hmc_sampler():
    g = GibbsSampler(MockBinfPDF(), BinfState({'x': 2.0, 'y': 3.0}),
                     {'x': MockSampler('x'),
                      'y': HMCSampler(MockBinfPDF(), 1.0, 0.1, 12)})
    g.state.update_variables(x=5.0)
    g.state.update_variables(y=2.3)
    g._update_subsampler_states()
    assert g.subsamplers['x'].state == 5.0
    assert g.subsamplers['y'].state == 2.3


def test_gibbs_update_state():
    g = make_gibbs()
    g._update_state(x=34.0)
    assert g.state.variables['x'] == 34.0


def test_gibbs_sweep_known_answer():
    """Alphabetical order + parameter refresh between sub-steps:
    x <- 0.5 * 2 * y(=3) = 3.0, then y <- 3.0 * 2 * x(=3.0) = 18.0."""
    g = make_gibbs()
    g._update_state(x=0.5)
    sample = g.sample()
    assert sample.variables == g.state.variables
    assert sample.variables['x'] == 3.0
    assert sample.variables['y'] == 18.0


def test_gibbs_stats():
    g = make_gibbs()
    stats = g.last_draw_stats
    assert set(stats) == {'x', 'y'}
    assert 'testlastdrawstats2.0' in stats['x']
    assert 'testlastdrawstats3.0' in stats['y']
    ss = g.sampling_stats
    assert 'testsamplingstats2.0' in ss and 'testsamplingstats3.0' in ss


def test_gibbs_checkstate():
    with pytest.raises(TypeError):
        make_gibbs()._checkstate([1, 2])


def test_example_models_and_samplers_have_no_cpu_evaluation_path():
    """The package's own model classes compute in HIP kernels only: host
    (numpy) values are refused with a TypeError instead of being evaluated with
    numpy.  (User plug-ins like the mocks above compute however they like.)"""
    from binf_amd.example.distance import DistanceForwardModel
    from binf_amd.example.likelihood import POLYVAL, ForwardModel, GaussianErrorModel
    from binf_amd.example.priors import GaussianPrior
    from binf_amd.example.samplers import RWMCSampler
    np = numpy
    xs, ys = np.linspace(-1, 1, 5), np.zeros(5)
    with pytest.raises(TypeError, match='no CPU path'):
        ForwardModel(xs, POLYVAL)(coefficients=np.ones(3))
    with pytest.raises(TypeError, match='no CPU path'):
        ForwardModel(xs, POLYVAL).jacobi_matrix(coefficients=np.ones(3))
    with pytest.raises(TypeError, match='no CPU path'):
        GaussianErrorModel(ys).log_prob(mock_data=np.zeros(5), precision=1.0)
    with pytest.raises(TypeError, match='no CPU path'):
        GaussianErrorModel(ys).gradient(mock_data=np.zeros(5), precision=1.0)
    with pytest.raises(TypeError, match='no CPU path'):
        GaussianPrior(np.zeros(3), np.ones(3)).log_prob(coefficients=np.ones(3))
    with pytest.raises(TypeError, match='no CPU path'):
        DistanceForwardModel(4)(coordinates=np.zeros(12))
    with pytest.raises(TypeError, match='no CPU path'):
        RWMCSampler(None, np.ones(3), 0.1).sample()
    import torch
    with pytest.raises(TypeError, match='a cpu tensor'):
        GaussianErrorModel(ys).log_prob(mock_data=torch.zeros(5, dtype=torch.float64), precision=1.0)


def test_gibbs_not_applicable_hooks_exist():
    """gibbs.py:153-163: the single-chain-MC hooks a Gibbs sampler does not use."""
    g = make_gibbs()
    assert g._calc_pacc() is None and g._propose() is None


def test_density_leftovers_and_validation_hook():
    """binf/pdf/__init__.py:39-47 (estimator / estimate are not implemented in the
    reference either) and the parameter validation hook of binf/model/__init__.py:52-57."""
    pdf = MockBinfPDF()
    with pytest.raises(NotImplementedError):
        pdf.estimator
    pdf.estimator = 'anything'               # accepted and ignored, as in the reference
    with pytest.raises(NotImplementedError):
        pdf.estimate([1.0])
    seen = []

    class Picky(MockBinfPDF):
        def _validate(self, name, parameter):
            seen.append(name)
            if parameter.value == 13.0:
                raise ValueError('refused')
    p = Picky()                               # the constructor fills 'ParamA' once
    assert seen == ['ParamA']
    p['ParamA'] = Parameter(3.0, 'ParamA')
    assert seen == ['ParamA', 'ParamA'] and p['ParamA'].value == 3.0
    with pytest.raises(ValueError):
        p['ParamA'] = Parameter(13.0, 'ParamA')
    assert p['ParamA'].value == 3.0


def test_hmc_sampler_attributes_follow_the_reference_run():
    """``tests/golden/ref_hmc_attributes.json``: what the REFERENCE's own HMCSampler object answers
    for the attributes GibbsSampler and user code read (hmc.py:56-90,127-134,166-181) -- produced by
    oracle/gen_ref_leapfrog.py from the reference's source (csb import dropped, no stand-in).  This
    package's HMCSampler gives the same answers (no kernel involved; host tensors)."""
    import json
    import os
    import torch
    from binf_amd.pdf import IsotropicGaussian
    from binf_amd.samplers.hmc import HMCSampler
    from conftest import GOLDEN_DIR
    ref = json.load(open(os.path.join(GOLDEN_DIR, 'ref_hmc_attributes.json')))
    assert 'hmc.py:56-90' in ref['provenance']
    state = torch.arange(3.0, dtype=torch.float64)
    s = HMCSampler(IsotropicGaussian(), state, 0.25, 7)                 # variable_name left at None
    f = ref['fresh']
    for name in ('acceptance_rate', 'variable_name', 'last_move_accepted', 'n_accepted', 'counter',
                 'timestep', 'nsteps', 'timestep_adaption_limit', 'adaption_uprate', 'adaption_downrate'):
        assert getattr(s, name) == f[name], name
    stats = s.last_draw_stats
    assert list(stats) == list(f['last_draw_stats'])
    got = stats['HMC']
    assert list(got._fields) == f['last_draw_stats']['HMC']['fields']
    assert list(got) == f['last_draw_stats']['HMC']['values']
    s.n_accepted, s.counter = 3, 4
    s._last_move_accepted = True
    a = ref['after_3_of_4']
    assert s.acceptance_rate == a['acceptance_rate'] and s.last_move_accepted is a['last_move_accepted']
    assert list(s.last_draw_stats['HMC']) == a['last_draw_stats_values']
    named = HMCSampler(IsotropicGaussian(), state, 0.25, 7, variable_name='coefficients')
    assert named.variable_name == ref['named']['variable_name']
    assert list(named.last_draw_stats) == ref['named']['last_draw_stats_keys']
    c = s._copy_state(s.state)
    assert bool(torch.equal(c, s.state)) is ref['copy_state']['equal']
    assert (c is s.state) is ref['copy_state']['same_object']
    # quirk Q1, as the reference: sample() with variable_name None fails with a TypeError
    with pytest.raises(TypeError):
        s.sample()


def test_binf_state_follows_the_reference_run():
    """BinfState (binf/samplers/__init__.py:9-57) executed from the reference's source (its two csb
    import statements dropped): `.variables` is a COPY, updates merge, momenta likewise -- this
    package's BinfState answers the same."""
    import json
    import os
    from binf_amd.samplers import BinfState
    from conftest import GOLDEN_DIR
    ref = json.load(open(os.path.join(GOLDEN_DIR, 'ref_hmc_attributes.json')))['binf_state']
    st = BinfState({'b': 2.0, 'a': 1.0})
    view = st.variables
    view['a'] = 99.0
    st.update_variables(c=3.0, a=1.5)
    st.update_momenta(a=-1.0)
    assert dict(st.variables) == ref['variables_after_update']
    assert (st.variables['a'] == 1.5) is ref['copy_is_detached']
    assert view == ref['view_after_write'] and dict(st.momenta) == ref['momenta']
    assert (BinfState().variables == {} and BinfState().momenta == {}) is ref['fresh_is_empty']


def test_mirror_offers_every_name_the_reference_modules_define():
    """``tests/golden/ref_api_surface.json`` (oracle/gen_ref_surface.py): every class, function and
    method name of the reference's hot-path modules and of its example application.  After
    ``s/binf/binf_amd/`` an import of any of them succeeds and every method is there (inherited
    or defined; the private ones too -- subclasses written against the reference call them)."""
    import importlib
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_api_surface.json')
    surface = json.load(open(path))['modules']
    assert len(surface) == 15
    missing = []
    for mod, names in surface.items():
        m = importlib.import_module(mod.replace('binf', 'binf_amd', 1))
        for name, methods in names.items():
            if not hasattr(m, name):
                missing.append('%s.%s' % (mod, name))
                continue
            for meth in methods or []:
                if not hasattr(getattr(m, name), meth):
                    missing.append('%s.%s.%s' % (mod, name, meth))
    assert missing == []

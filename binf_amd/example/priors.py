"""
Priors of the example application (mirror of reference
``binf/example/priors.py``).  Per-chain scalars (``precision``) are ``[C]``
tensors or Python floats; ``coefficients`` is ``[C x K]``.
"""
import numpy as np
import torch

from binf_amd import ArrayParameter, _native
from binf_amd.params import Parameter as ScalarParameter
from binf_amd.params import ParameterNotFoundError
from binf_amd.pdf.priors import AbstractPrior


class GammaPrior(AbstractPrior):
    """log p(precision) = (shape-1)*log(precision) - precision*rate
    (reference ``:10-32``)."""

    def __init__(self, shape, rate):
        super(GammaPrior, self).__init__('precision_prior')
        self.shape = shape
        self.rate = rate
        self._register_variable('precision')
        self.update_var_param_types(precision=ScalarParameter)
        self._set_original_variables()

    def _evaluate_log_prob(self, precision):
        if isinstance(precision, torch.Tensor):
            # per-chain precisions: one launch, the roundings of the expression below
            _native.require_device(precision, 'precision')
            return _native.gamma_logp(precision.to(torch.float64).contiguous(),
                                      self.shape, self.rate)
        # one precision for all chains (a Python / numpy scalar): host arithmetic
        return (self.shape - 1.0) * np.log(precision) - precision * self.rate

    def clone(self):
        # Reference quirk Q6, kept: the copy is built with (shape, shape), so
        # the RATE of a cloned / conditional GammaPrior equals its shape
        # (reference :27-32).  Conditional posteriors -- the ones the Gibbs
        # subsamplers see -- therefore use rate = shape.
        copy = self.__class__(self.shape, self.shape)
        copy.set_fixed_variables_from_pdf(self)
        return copy


class GaussianPrior(AbstractPrior):
    """log p(coefficients) = -0.5*sum((c-means)**2/variances) (reference
    ``:35-64``).  ``coefficients`` is registered NON-differentiable, as in the
    reference: the posterior's force skips this prior (quirk Q4)."""

    def __init__(self, means, variances):
        super(GaussianPrior, self).__init__('coefficients_prior')
        self._register('means')
        self._register('variances')
        self['means'] = ArrayParameter(means, 'means')
        self['variances'] = ArrayParameter(variances, 'variances')
        self._dev = {}
        self._register_variable('coefficients')
        self.update_var_param_types(coefficients=ArrayParameter)
        self._set_original_variables()

    def _vec(self, name, device):
        key = (name, device)
        v = self[name].value
        if key not in self._dev or self._dev[key][0] is not v:
            t = v if isinstance(v, torch.Tensor) else \
                torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64))
            self._dev[key] = (v, t.to(device))
        return self._dev[key][1]

    def _evaluate_log_prob(self, coefficients):
        _native.require_device(coefficients, 'coefficients')
        c2 = coefficients if coefficients.dim() == 2 else coefficients.reshape(1, -1)
        dev = c2.device
        return _native.row_sumsq_diff(c2.contiguous(), self._vec('means', dev),
                                      scale=-0.5, weights=self._vec('variances', dev))

    def _evaluate_gradient(self, **variables):
        # the reference's implementation reads parameters that do not exist
        # ('mu', 'sigma', reference :56-60) and is never reached (quirk Q4)
        raise ParameterNotFoundError('mu')

    def clone(self):
        return self.__class__(self['means'].value, self['variances'].value)


def make_priors():
    PP = GammaPrior(1.0, 0.2)
    CP = GaussianPrior(means=np.array([0.0, 0.0, 0.0, 0.0]),
                       variances=np.ones(4) * 5)
    return {PP.name: PP, CP.name: CP}

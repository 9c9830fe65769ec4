"""
The example application's likelihood pieces: polynomial forward model and
Gaussian error model (mirror of reference ``binf/example/likelihood.py``),
chain-batched and backed by the HIP kernels of ``csrc/poly.hip``.
"""
import numpy as np
import torch

from binf_amd import ArrayParameter, _native
from binf_amd.model.errormodels import AbstractErrorModel
from binf_amd.model.forwardmodels import AbstractForwardModel
from binf_amd.params import Parameter as ScalarParameter

POLYVAL = np.polynomial.polynomial.polyval


def _as2d(x):
    return x if x.dim() == 2 else x.reshape(1, -1)


def _unchanged(obj, base, names):
    """True if ``obj``'s class still uses ``base``'s implementation of every
    method in ``names`` -- a subclass that overrides one of them must be
    evaluated as written, not through the fused kernels of the base model."""
    return all(getattr(type(obj), n, None) is getattr(base, n) for n in names)


class ForwardModel(AbstractForwardModel):
    """mock = polynomial(xses, coefficients); Jacobian rows are the powers of
    ``xses`` (reference ``:11-37``).

    ``polynomial`` is the callable the reference takes
    (``numpy.polynomial.polynomial.polyval`` in ``example_script.py:21``).  For
    exactly that callable the Horner recurrence runs natively (bit-identical
    to numpy's); any other callable is applied as given.

    The design matrix ``vstack([xses**i])`` is what the reference rebuilds on
    every gradient call (``:28-30``); here it is built once per coefficient
    count -- on the host with the same numpy expression, so its entries carry
    the reference's bits -- and kept in HBM.
    """

    def __init__(self, xses, polynomial):
        super(ForwardModel, self).__init__('polynomial')
        self._dev = {}
        self.xses = xses
        self.polynomial = polynomial
        self._register_variable('coefficients', differentiable=True)
        self.update_var_param_types(coefficients=ArrayParameter)
        self._set_original_variables()

    # -- device-resident model data -----------------------------------------
    # The device copies are derived from ``xses`` when first needed and shared
    # with clones; assigning a new ``xses`` drops THIS object's copies (the
    # reference reads the attribute on every call).  Mutating the array in place
    # after first use is not seen: model data are immutable once evaluated.
    @property
    def xses(self):
        return self._xses

    @xses.setter
    def xses(self, value):
        self._xses = value
        self._dev = {}

    def _xs_host(self):
        x = self.xses
        return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) \
            else np.asarray(x, dtype=np.float64)

    def xs_device(self, device):
        key = ('xs', device)
        if key not in self._dev:
            self._dev[key] = torch.from_numpy(
                np.ascontiguousarray(self._xs_host())).to(device)
        return self._dev[key]

    def design_matrix(self, n_coefficients, device):
        key = ('A', int(n_coefficients), device)
        if key not in self._dev:
            xs = self._xs_host()
            A = np.vstack([xs ** i for i in range(int(n_coefficients))])
            self._dev[key] = torch.from_numpy(np.ascontiguousarray(A)).to(device)
        return self._dev[key]

    @property
    def is_native(self):
        return self.polynomial is POLYVAL

    # -- model interface ------------------------------------------------------
    def _evaluate(self, coefficients):
        if not self.is_native:
            # a user-supplied polynomial callable is applied as given
            return self.polynomial(self.xses, coefficients)
        _native.require_device(coefficients, 'coefficients')
        out = _native.poly_forward(_as2d(coefficients),
                                   self.xs_device(coefficients.device))
        return out if coefficients.dim() == 2 else out.reshape(-1)

    def _evaluate_jacobi_matrix(self, coefficients):
        _native.require_device(coefficients, 'coefficients')
        return self.design_matrix(coefficients.shape[-1], coefficients.device)

    def clone(self):
        copy = self.__class__(self.xses, self.polynomial)
        copy._dev = self._dev            # immutable model data: share it
        self._set_parameters(copy)
        return copy

    def native_spec(self):
        if self.is_native and _unchanged(self, ForwardModel,
                                         ('_evaluate', '_evaluate_jacobi_matrix')):
            return ('polynomial', self)
        return None


class GaussianErrorModel(AbstractErrorModel):
    """log p = -0.5*sum((mock-ys)**2)*precision + len(ys)*0.5*log(precision);
    gradient w.r.t. mock_data = (mock-ys)*precision (reference ``:40-68``)."""

    def __init__(self, ys):
        super(GaussianErrorModel, self).__init__('error_model')
        self._dev = {}
        self.ys = ys
        self._register_variable('mock_data')
        self._register_variable('precision')
        self.update_var_param_types(mock_data=ArrayParameter,
                                    precision=ScalarParameter)
        self._set_original_variables()

    @property
    def ys(self):
        return self._ys

    @ys.setter
    def ys(self, value):
        # new data: drop this object's device copies (clones keep theirs)
        self._ys = value
        self._dev = {}
        if hasattr(self, '_ymat'):
            self._ymat = {}

    def ys_device(self, device):
        if device not in self._dev:
            y = self.ys
            y = y.detach().cpu().numpy() if isinstance(y, torch.Tensor) \
                else np.asarray(y, dtype=np.float64)
            self._dev[device] = torch.from_numpy(np.ascontiguousarray(y)).to(device)
        return self._dev[device]

    def _evaluate_log_prob(self, mock_data, precision):
        _native.require_device(mock_data, 'mock_data')
        return _native.gauss_err_logp(_as2d(mock_data),
                                      self.ys_device(mock_data.device), precision)

    def _evaluate_gradient(self, mock_data, precision):
        _native.require_device(mock_data, 'mock_data')
        out = _native.gauss_err_grad(_as2d(mock_data),
                                     self.ys_device(mock_data.device), precision)
        return out if mock_data.dim() == 2 else out.reshape(-1)

    def clone(self):
        copy = self.__class__(self.ys)
        copy._dev = self._dev
        copy.set_fixed_variables_from_pdf(self)
        return copy

    def native_spec(self):
        if _unchanged(self, GaussianErrorModel, ('_evaluate_log_prob', '_evaluate_gradient')):
            return ('gaussian', self)
        return None


def make_likelihood(xses, ys, polynomial):
    from binf_amd.pdf.likelihoods import Likelihood
    return Likelihood('points', ForwardModel(xses, polynomial),
                      GaussianErrorModel(ys))


# the fused kernels of this module's models: importing the models registers their kind
from binf_amd.example import native_poly as _polynomial_kind  # noqa: E402,F401

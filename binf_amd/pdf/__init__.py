"""
Probability-density base class and the isotropic Gaussian.

Mirror of reference ``binf/pdf/__init__.py`` (AbstractBinfPDF ``:19-160``,
TestHO ``:163-191``) with chain-batched values: ``log_prob`` returns one value
per chain (``[C]`` tensor) and ``gradient`` a ``[C x D]`` tensor.  As in the
reference, ``gradient`` is the gradient of the ENERGY, -log p.
"""
from binf_amd import AbstractBinfNamedCallable
from binf_amd.params import (ArrayParameter, Parameter, ParameterHolder,  # noqa
                             ParameterNotFoundError)


class AbstractBinfPDF(ParameterHolder, AbstractBinfNamedCallable):
    """A density over named variables with named parameters.

    ``log_prob(**vars)`` / ``gradient(**vars)`` first inject the values of
    fixed variables (``_complete_variables``) and then call the subclass's
    ``_evaluate_log_prob`` / ``_evaluate_gradient`` (reference
    ``binf/pdf/__init__.py:91-111,153-160``).  They deliberately do not check
    the argument count, like the reference.
    """

    def __init__(self, name='', **args):
        ParameterHolder.__init__(self)
        AbstractBinfNamedCallable.__init__(self, name)

    # -- csb.statistics.pdf.AbstractDensity leftovers the reference keeps -------
    @property
    def estimator(self):
        raise NotImplementedError          # reference binf/pdf/__init__.py:39-41

    @estimator.setter
    def estimator(self, strategy):
        pass                               # :42-44

    def estimate(self, data):
        raise NotImplementedError          # :46-47

    # -- evaluation ---------------------------------------------------------
    def _evaluate_log_prob(self, **variables):
        raise NotImplementedError

    def log_prob(self, **variables):
        self._complete_variables(variables)
        return self._evaluate_log_prob(**variables)

    def gradient(self, **variables):
        self._complete_variables(variables)
        return self._evaluate_gradient(**variables)

    def _evaluate(self, **variables):
        # exp(log_prob), reference :87-89 (csb's clipped exp)
        lp = self.log_prob(**variables)
        try:
            import torch
        except ImportError:  # pragma: no cover
            torch = None
        if torch is not None and isinstance(lp, torch.Tensor) and lp.is_cuda:
            from binf_amd import _native
            return _native.clipped_exp(lp)
        # host values only come from user plug-ins that compute on the host
        import numpy
        return numpy.exp(numpy.clip(lp, -308.0, 709.0))

    def _complete_variables(self, variables):
        for p in self.parameters:
            if p in self._original_variables:
                variables[p] = self[p].value

    # -- conditioning -------------------------------------------------------
    def clone(self):
        raise NotImplementedError

    def conditional_factory(self, **fixed_vars):
        """A copy with the given variables fixed (names that are not variables
        of this PDF are ignored) -- reference :49-70."""
        result = self.clone()
        result.fix_variables(**self._get_variables_intersection(fixed_vars))
        return result

    def set_fixed_variables_from_pdf(self, pdf):
        """Fix here whatever ``pdf`` has fixed that this object still treats
        as a variable (reference :142-151)."""
        have = set(self.parameters)
        theirs = {p: pdf[p].value for p in pdf.parameters if p not in have}
        self.fix_variables(**self._get_variables_intersection(theirs))

    # -- native dispatch ----------------------------------------------------
    def native_hmc_spec(self, variable_name):
        """Descriptor of a fused HIP trajectory kernel that samples
        ``variable_name`` from this PDF, or None (generic per-step tier)."""
        return None


class IsotropicGaussian(AbstractBinfPDF):
    """log p(x) = -0.5 * k * sum((x - x0)**2) -- the reference's own Gaussian,
    ``TestHO`` (``binf/pdf/__init__.py:163-191``; uninstantiable there because
    it imports a missing package).  Parameters ``k`` and ``x0`` are scalars,
    the variable is ``x`` (configurable)."""

    def __init__(self, k=1.0, x0=0.0, name='TestHO', variable_name='x'):
        super(IsotropicGaussian, self).__init__(name=name)
        self._register('k')
        self._register('x0')
        self['k'] = Parameter(k, name='k')
        self['x0'] = Parameter(x0, name='x0')
        self._vname = variable_name
        self._register_variable(variable_name, differentiable=True)
        self._set_original_variables()
        self.update_var_param_types(**{variable_name: ArrayParameter})

    def _evaluate_log_prob(self, **variables):
        from binf_amd import _native
        x = variables[self._vname]
        k, x0 = self['k'].value, self['x0'].value
        # (-0.5 * k) * np.sum((x - x0) ** 2), numpy order  (reference :185)
        return _native.row_sum(_as2d(x), _native.ROW_SUMSQ_SHIFT, shift=x0,
                               scale=-0.5 * k)

    def _evaluate_gradient(self, **variables):
        from binf_amd import _native
        x = variables[self._vname]
        out = _native.gauss_grad(_as2d(x), self['k'].value, self['x0'].value)
        return out.view(x.shape)

    def clone(self):
        copy = self.__class__(self['k'].value, self['x0'].value, self.name,
                              self._vname)
        copy.set_fixed_variables_from_pdf(self)
        return copy

    def native_hmc_spec(self, variable_name):
        if variable_name == self._vname and variable_name in self.variables:
            return ('gauss', float(self['k'].value), float(self['x0'].value))
        return None


# the reference's name for it (``from binf.pdf import TestHO``)
TestHO = IsotropicGaussian


def _as2d(x):
    return x if x.dim() == 2 else x.reshape(1, -1)


# the fused kernels of the Gaussian register themselves as the kind 'gauss' (binf_amd/native.py)
from binf_amd.pdf import native_gauss  # noqa: E402,F401

"""
Forward models: theta -> idealised ("mock") data, plus the Jacobian of that
map.  Mirror of reference ``binf/model/forwardmodels.py:10-66``.

Batched convention: ``fwm(**vars)`` returns ``[C x n_data]`` for ``[C x
n_params]`` inputs; ``jacobi_matrix`` returns ``[n_params x n_data]`` when it
does not depend on the chain (linear models) or ``[C x n_params x n_data]``.
"""
from binf_amd.model import AbstractModel


class AbstractForwardModel(AbstractModel):

    def __init__(self, name, parameters=()):
        super(AbstractForwardModel, self).__init__(name, parameters)

    @property
    def data(self):
        return self._data

    def jacobi_matrix(self, **variables):
        self._complete_variables(variables)
        return self._evaluate_jacobi_matrix(**variables)

    def _evaluate_jacobi_matrix(self, **model_parameters):
        self._check_differentiability(**model_parameters)

    def clone(self):
        raise NotImplementedError

    def _set_parameters(self, copy):
        """Give ``copy`` every parameter of this model it does not have yet;
        a name that was still a variable there becomes fixed (reference
        :59-66)."""
        for p in self.parameters:
            if p not in copy.parameters:
                copy._register(p)
                copy[p] = self[p].__class__(self[p].value, p)
                if p in copy.variables:
                    copy._delete_variable(p)

    def native_spec(self):
        """Descriptor of a HIP implementation of this model, or None."""
        return None

"""
Models: named callables that hold parameters (forward models, and the base the
error models share their parameter handling with).  Mirror of reference
``binf/model/__init__.py:13-91``.
"""
from binf_amd import AbstractBinfNamedCallable
from binf_amd.params import ParameterHolder, ParameterNotFoundError  # noqa: F401


class AbstractModel(ParameterHolder, AbstractBinfNamedCallable):

    def __init__(self, name, parameters=()):
        ParameterHolder.__init__(self)
        AbstractBinfNamedCallable.__init__(self, name)
        for prm in parameters:
            self._register(prm.name)
            self[prm.name] = prm

    def _complete_variables(self, variables):
        # inject the values of variables that have been fixed (reference :78-85)
        for p in self.parameters:
            if p in self._original_variables:
                variables[p] = self[p].value

    def _reduce_variables(self, **variables):
        for p in self.parameters:
            variables.pop(p, None)
        return variables

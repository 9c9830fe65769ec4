"""
Sampler state container (mirror of reference ``binf/samplers/__init__.py:9-57``).

Variable values are chain-batched device tensors (``[C x D]`` vectors, ``[C]``
per-chain scalars); the container itself is unchanged in behaviour.
"""


class BinfState(object):
    """Named variable values of a Markov-chain state.

    ``variables`` hands out a shallow copy of the name -> value mapping (the
    values themselves are shared), exactly like the reference (``:26-34``).
    ``momenta`` exists for interface parity and is unused there as well.
    """

    def __init__(self, variables=None, momenta=None):
        self._variables = {}
        self._momenta = {}
        self.update_variables(**(variables or {}))
        self.update_momenta(**(momenta or {}))

    @property
    def variables(self):
        return dict(self._variables)

    def update_variables(self, **variables):
        self._variables.update(variables)

    @property
    def momenta(self):
        return dict(self._momenta)

    def update_momenta(self, **momenta):
        self._momenta.update(momenta)

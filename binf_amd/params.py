"""
Self-contained parameter objects and parameter-holding mix-in.

The reference takes these from the third-party CSB toolbox
(``csb.statistics.pdf.parameterized``: AbstractParameter / Parameter /
ParameterizedDensity -- see reference ``binf/__init__.py:13``,
``binf/pdf/__init__.py:11``).  Only the behaviour binf relies on is provided:

* ``p.value`` / ``p.set(v)`` / ``p.name``;
* ``child.bind_to(parent)``: the child follows the parent -- a later
  ``parent.set(v)`` pushes ``v`` into every bound child, transitively
  (used at ``binf/pdf/posteriors.py:53-55``, ``binf/pdf/likelihoods.py:84-88``
  and relied on by ``binf/samplers/gibbs.py:62``);
* a holder with ``_register(name)``, ``holder[name]``, ``holder.parameters``,
  ``get_params()`` (``binf/model/__init__.py:31-76`` has the same surface for
  models).

Values may be Python scalars, numpy arrays or torch tensors ([C] per-chain
scalars, [C x D] per-chain vectors); nothing here touches their contents.
"""
from collections import OrderedDict


class ParameterValueError(ValueError):

    def __init__(self, name, value):
        super(ParameterValueError, self).__init__(
            'invalid value for parameter %r: %r' % (name, value))
        self.name = name
        self.value = value


class ParameterNotFoundError(AttributeError):
    """Raised on access to a parameter name that was never registered
    (reference: ``binf/pdf/__init__.py:14-16``)."""


class AbstractParameter(object):

    def __init__(self, value=None, name=None, base=None):
        self._name = name
        self._followers = []
        self._leader = None
        self._value = None
        if value is not None:
            self._value = self._validate(value)
        if base is not None:
            self.bind_to(base)

    def _validate(self, value):
        return value

    @property
    def name(self):
        return self._name

    @property
    def value(self):
        return self._value

    def set(self, value):
        """Assign a new value and push it to every parameter bound to this
        one."""
        self._value = self._validate(value)
        for f in self._followers:
            f.set(self._value)

    def bind_to(self, leader):
        """Make this parameter follow ``leader``."""
        if leader is self:
            raise ValueError('a parameter cannot be bound to itself')
        if self._leader is not None and self in self._leader._followers:
            self._leader._followers.remove(self)
        self._leader = leader
        if leader is not None and self not in leader._followers:
            leader._followers.append(self)

    def __repr__(self):
        return '<%s %s=%r>' % (self.__class__.__name__, self._name, self._value)


class Parameter(AbstractParameter):
    """Scalar parameter.  Python numbers are stored as float (csb's Parameter
    does the same); tensors / arrays holding one value per chain pass through."""

    def _validate(self, value):
        if isinstance(value, (int, float)) and not isinstance(value, bool):
            return float(value)
        return value


class ArrayParameter(AbstractParameter):
    """Array-valued parameter (reference ``binf/__init__.py:238-244`` wraps
    the value in ``numpy.array``; device tensors are kept as they are)."""

    def _validate(self, value):
        try:
            import torch
            if isinstance(value, torch.Tensor):
                return value
        except ImportError:  # pragma: no cover
            pass
        import numpy
        try:
            return numpy.array(value)
        except (TypeError, ValueError):
            raise ParameterValueError(self.name, value)


class ParameterHolder(object):
    """Named parameter slots: ``_register`` declares a slot, item access
    reads / fills it."""

    def __init__(self):
        self._params = OrderedDict()

    def _register(self, name):
        if name not in self._params:
            self._params[name] = None

    def __getitem__(self, name):
        if name in self._params:
            return self._params[name]
        raise ParameterNotFoundError(name)

    def __setitem__(self, name, parameter):
        if name not in self._params:
            raise ParameterNotFoundError(name)
        if not isinstance(parameter, AbstractParameter):
            raise TypeError('%r is not a parameter object' % (parameter,))
        self._validate(name, parameter)
        self._params[name] = parameter

    def _validate(self, name, parameter):
        """Parameter validation hook, called before a slot is filled (reference
        ``binf/model/__init__.py:52-57``; subclasses raise to refuse a value)."""

    @property
    def parameters(self):
        return tuple(self._params)

    def get_params(self):
        return [self._params[n] for n in self._params]

    def set_params(self, *values, **named):
        for n, v in zip(self.parameters, values):
            self[n].set(v.value if isinstance(v, AbstractParameter) else v)
        for n, v in named.items():
            self[n].set(v.value if isinstance(v, AbstractParameter) else v)

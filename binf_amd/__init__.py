"""
binf_amd -- MI355X-native engine behind binf's HMC hot path.

Host-side mirror of the reference's plug-in surface (same class names, keyword
calling convention and error behaviour) with chain-batched values: where the
reference passes a ``[D]`` numpy array, this package passes a ``[C x D]`` fp64
ROCm tensor, and per-chain scalars are ``[C]`` tensors.  The arithmetic of the
hot path runs in hand-written HIP kernels (``binf_amd/csrc``) reached through
the C ABI in ``include/binf_hip.h``.

This module: the named-callable core (reference ``binf/__init__.py:16-226``).
"""
__version__ = '0.1.0'

from binf_amd.params import (AbstractParameter, ArrayParameter,  # noqa: F401
                             Parameter, ParameterNotFoundError,
                             ParameterValueError)


class AbstractBinfNamedCallable(object):
    """A function of named variables.

    Subclasses register their variable names once; callers then pass values
    by keyword.  A variable can later be *fixed*: it leaves the variable set
    and becomes a parameter of the object, whose value is injected into every
    evaluation (reference ``binf/__init__.py:34-61,105-120,160-179,209-226``).

    Subclasses provide ``_evaluate`` (and optionally ``_evaluate_gradient``),
    ``_complete_variables`` and the parameter slots (``_register`` /
    item access), normally via :class:`binf_amd.params.ParameterHolder`.
    """

    def __init__(self, name):
        self._name = name
        self._variables = set()
        self._differentiable_variables = set()
        self._var_param_types = {}
        self._original_variables = set()

    # -- variable registry --------------------------------------------------
    @property
    def name(self):
        return self._name

    @property
    def variables(self):
        return self._variables

    @property
    def differentiable_variables(self):
        return self._differentiable_variables

    @property
    def var_param_types(self):
        return dict(self._var_param_types)

    def update_var_param_types(self, **types):
        self._var_param_types.update(types)

    def _set_original_variables(self):
        self._original_variables.update(self._variables)

    def _register_variable(self, name, differentiable=False):
        if type(name) != str:
            raise ValueError('Variable name must be a string, not %s'
                             % type(name).__name__)
        if name in self._variables:
            raise ValueError('Variable name "%s" must be unique' % name)
        self._variables.add(name)
        if differentiable:
            self._differentiable_variables.add(name)

    def _delete_variable(self, name):
        if name not in self._variables:
            raise ValueError('"%s": unknown variable name' % name)
        self._variables.discard(name)
        self._differentiable_variables.discard(name)

    def _get_variables_intersection(self, candidates):
        return {k: v for k, v in candidates.items() if k in self._variables}

    # -- evaluation ---------------------------------------------------------
    def _check_arity(self, variables):
        if len(variables) != len(self._variables):
            raise ValueError('Function called with %d arguments instead of %d!'
                             % (len(variables), len(self._variables)))

    def __call__(self, **variables):
        self._check_arity(variables)
        self._complete_variables(variables)
        return self._evaluate(**variables)

    def gradient(self, **variables):
        self._check_arity(variables)
        self._complete_variables(variables)
        return self._evaluate_gradient(**variables)

    def _evaluate(self, **variables):
        raise NotImplementedError

    def _evaluate_gradient(self, **variables):
        raise NotImplementedError

    def _complete_variables(self, variables):
        raise NotImplementedError

    def _check_differentiability(self, **variables):
        if not (set(variables) & self._differentiable_variables):
            raise ValueError('Function cannot be differentiated w.r.t. any of '
                             'the variables %s' % sorted(variables))

    # -- fixing variables ---------------------------------------------------
    def fix_variables(self, **fixed_vars):
        """Turn variables into parameters holding the given values.  Unknown
        names and names without a declared parameter type raise ValueError
        (pinned by reference ``binf/tests/pdf/__init__.py:57``)."""
        for v, value in fixed_vars.items():
            if v not in self._variables:
                raise ValueError('%r is not a variable of %r' % (v, self))
            self._delete_variable(v)
            self._register(v)
            if v not in self._var_param_types:
                raise ValueError('Parameter type for variable "%s" not '
                                 'defined' % v)
            self[v] = self._var_param_types[v](value, v)

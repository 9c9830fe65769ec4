"""
Posterior = product of likelihoods and priors.  Mirror of reference
``binf/pdf/posteriors.py:15-210``; this is the gradient dispatch every
leapfrog step of the generic HMC tier goes through.

Component order.  The reference iterates Python-2 dicts (hash order; for the
log-prob even object-address order, ``:123,139-145``), so the order in which
it adds component terms is not reproducible between processes.  This build
fixes the order to SORTED COMPONENT NAME for both ``log_prob`` and
``gradient``; single-component cases are unaffected.

Quirk kept on purpose (Q4): a component contributes to ``gradient`` only if it
has at least one variable AND at least one differentiable variable
(``:183``) -- priors that register their variable as non-differentiable are
part of the energy but not of the force.
"""
from binf_amd.pdf import AbstractBinfPDF


SUM_TERMS_MAX = 16        # terms of one binf_sum_terms launch (include/binf_hip.h)


def _sum_in_order(terms):
    """``((t0 + t1) + t2) + ...``: per-chain device tensors in launches of
    ``binf_sum_terms_bcast_f64`` (one launch for up to 16 terms, the running sum
    carried into the next launch beyond that: the same left-to-right order);
    Python floats ride along as scalars, 0-dim device tensors as broadcast device
    scalars (never read back to the host).  Host values (the reference's own
    unit-test cases) are added with Python's ``+``.  There is no PyTorch
    arithmetic here: device terms the kernel cannot take -- host tensors mixed
    with device tensors, other dtypes, non-contiguous or differently shaped
    vectors -- are refused with a ``TypeError`` / ``ValueError``."""
    if len(terms) == 1:
        return terms[0]
    try:
        import torch
    except ImportError:  # pragma: no cover
        torch = None

    def tensor(t):
        return torch is not None and isinstance(t, torch.Tensor)

    if not any(tensor(t) for t in terms):
        total = terms[0]
        for t in terms[1:]:
            total = total + t
        return total
    vectors = [t for t in terms if tensor(t) and t.dim() > 0]
    if not vectors:
        # only 0-dim tensors and floats: a one-element "vector" carries the sum
        first = next(t for t in terms if tensor(t))
        vectors = [first.reshape(1)]
        terms = [t.reshape(1) if t is first else t for t in terms]
        squeeze = True
    else:
        squeeze = False
    ref = vectors[0]
    for t in terms:
        if not tensor(t):
            continue
        if not t.is_cuda or t.dtype != torch.float64:
            raise TypeError('Posterior: component terms must be fp64 device tensors (or Python '
                            'floats), got %s on %s' % (t.dtype, t.device))
        if t.dim() > 0 and (tuple(t.shape) != tuple(ref.shape) or not t.is_contiguous()):
            raise ValueError('Posterior: component terms must be contiguous and of one shape, got '
                             '%s (contiguous: %s) beside %s' % (tuple(t.shape), t.is_contiguous(),
                                                                tuple(ref.shape)))
    from binf_amd import _native
    total = None
    rest = list(terms)
    while rest:
        take = SUM_TERMS_MAX - (0 if total is None else 1)
        chunk, rest = rest[:take], rest[take:]
        total = _native.sum_terms(([] if total is None else [total]) + chunk, like=ref)
    return total.reshape(()) if squeeze else total


def _zero_force(values, all_values=()):
    """The reference's ``numpy.zeros(sum(len(v) for differentiable v passed))``
    (``posteriors.py:177-180``) for chain-batched values: device tensors ``[C x d_i]``
    (or ``[d_i]``, one chain) give a zero tensor ``[C x sum d_i]`` (``[sum d_i]``) on
    their device -- the shape a gradient of those variables has; host values give the
    reference's numpy vector.  With NO differentiable variable among those passed the
    reference's vector is EMPTY (and its ``HMCSampler._leapfrog`` then fails to
    broadcast it, ``hmc.py:116``): batched, that is ``[C x 0]``."""
    try:
        import torch
    except ImportError:  # pragma: no cover
        torch = None
    is_t = lambda v: torch is not None and isinstance(v, torch.Tensor)
    tens = [v for v in values if is_t(v)]
    if not values:
        like = next((v for v in all_values if is_t(v)), None)
        if like is None:
            import numpy
            return numpy.zeros(0)
        lead = tuple(like.shape[:-1]) if like.dim() > 1 else ()
        return torch.zeros(lead + (0,), dtype=like.dtype, device=like.device)
    if not tens:
        import numpy
        return numpy.zeros(sum(len(v) if hasattr(v, '__len__') else 1 for v in values))
    if len(tens) != len(values):
        raise TypeError('Posterior.gradient: device tensors and host values mixed')
    first = tens[0]
    if len(tens) == 1:
        return torch.zeros_like(first)
    if any(t.dim() != first.dim() or t.shape[:-1] != first.shape[:-1] for t in tens):
        raise ValueError('Posterior.gradient: differentiable variables of different batch shape')
    width = sum(int(t.shape[-1]) if t.dim() > 0 else 1 for t in tens)
    return torch.zeros(tuple(first.shape[:-1]) + (width,), dtype=first.dtype, device=first.device)


class Posterior(AbstractBinfPDF):

    def __init__(self, likelihoods, priors, name='the one and only posterior'):
        super(Posterior, self).__init__(name)
        self._likelihoods = likelihoods
        self._priors = priors
        self._setup_parameters()
        self._components = dict(priors)
        self._components.update(likelihoods)
        self._register_component_variables(*self._get_component_variables())
        self._set_original_variables()

    # -- construction --------------------------------------------------------
    def _setup_parameters(self):
        """One posterior-level parameter per distinct component parameter
        name; the components' parameters follow it (``:44-55``)."""
        for group in (self._likelihoods, self._priors):
            for comp in group.values():
                for p in comp.parameters:
                    if p not in self.parameters:
                        self._register(p)
                        self[p] = comp[p].__class__(comp[p].value, comp[p].name)
                    comp[p].bind_to(self[p])

    def _get_component_variables(self):
        names, fixed, diff, types = [], [], [], []
        for comp in self._components.values():
            for v in comp.variables:
                names.append(v)
                types.append(comp.var_param_types[v])
                if v in comp.differentiable_variables:
                    diff.append(v)
            for p in comp.parameters:
                if p in comp._original_variables:
                    fixed.append(p)
        return names, set(fixed), set(diff), types

    def _register_component_variables(self, names, fixed, diff, types):
        for v in set(names):
            self._register_variable(str(v), differentiable=v in diff)
        self._original_variables.update(fixed)
        self.update_var_param_types(**dict(zip(names, types)))

    @property
    def likelihoods(self):
        return self._likelihoods

    @property
    def priors(self):
        return self._priors

    def _ordered_components(self):
        return [self._components[n] for n in sorted(self._components)]

    def _get_component_variables_list(self):
        return {c: c.variables for c in self._components.values()}

    # -- evaluation ----------------------------------------------------------
    def _evaluate_components(self, **model_parameters):
        return [c.log_prob(**{v: model_parameters[v] for v in c.variables})
                for c in self._ordered_components()]

    def _evaluate_log_prob(self, **model_parameters):
        # numpy.sum over a short list = sequential adds (reference :147-151)
        return _sum_in_order(self._evaluate_components(**model_parameters))

    def _evaluate_gradient(self, **variables):
        grads = []
        for f in self._ordered_components():
            if len(f.variables) > 0 and len(f.differentiable_variables) > 0:
                grads.append(f.gradient(**{x: variables[x] for x in variables
                                           if x in f.variables}))
        if not grads:
            # no differentiable component: the reference returns its zero vector, one
            # entry per element of the differentiable variables passed in (reference
            # :177-180) -- batched: [C x (sum of their widths)] on the variables' device
            diff = [variables[v] for v in variables if v in self.differentiable_variables]
            return _zero_force(diff, list(variables.values()))
        return _sum_in_order(grads)

    # -- copies --------------------------------------------------------------
    def clone(self):
        copy = self.__class__({n: c.clone() for n, c in self._likelihoods.items()},
                              {n: c.clone() for n, c in self._priors.items()},
                              self.name)
        copy.set_fixed_variables_from_pdf(self)
        return copy

    def conditional_factory(self, **fixed_vars):
        """A new posterior built from the conditional copies of every
        component (``:201-210``)."""
        liks = {n: c.conditional_factory(**fixed_vars)
                for n, c in self._likelihoods.items()}
        pris = {n: c.conditional_factory(**fixed_vars)
                for n, c in self._priors.items()}
        return self.__class__(liks, pris, self.name)

    # -- fused kernels: asked of the registry, never named here (binf_amd/native.py) --
    def native_hmc_spec(self, variable_name):
        """``(kind, *params)`` when this (conditional) posterior is one a registered
        kind integrates in a single launch per transition, else None."""
        from binf_amd import native
        return native.match('hmc', self, variable_name)

    def native_leapfrog_spec(self, variable_name):
        """``(kind, *params)`` of a fused leapfrog kernel that integrates
        ``variable_name`` under THIS posterior's force, or None."""
        from binf_amd import native
        return native.match('leapfrog', self, variable_name)

    def native_energy_spec(self, variable_name):
        """``(kind, *params)`` of a fused kernel for ``HMCSampler.sample()``'s energy
        ``0.5 * sum(p**2) - log_prob`` under THIS posterior, or None."""
        from binf_amd import native
        return native.match('energy', self, variable_name)

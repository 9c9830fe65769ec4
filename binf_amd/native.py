"""
Registry of fused kernels -- how a model gets a native fast path WITHOUT the core
knowing it.

The reference's value is that a user adds a model without touching the core
(``binf/pdf/__init__.py:19-160``: subclass ``AbstractBinfPDF``;
``binf/model/forwardmodels.py:10-66``: subclass ``AbstractForwardModel``).  The
chain-batched core keeps that property for fast paths too: ``HMCSampler``,
``Posterior``, ``Likelihood`` and ``GibbsSampler`` never name a model.  They ask
this registry, and a model module *registers a kind* when it is imported
(``binf_amd.example.native_poly``, ``binf_amd.example.distance``; the isotropic
Gaussian of ``binf_amd.pdf`` registers the built-in kind ``'gauss'``).

A **kind** is a name plus hooks, all optional:

recognition -- called with a PDF, answer ``params`` (any tuple) or None.  The core
hands ``(kind, *params)`` around as the *spec*; a PDF may equally return such a spec
itself from ``native_hmc_spec(name)`` / ``native_leapfrog_spec(name)`` /
``native_energy_spec(name)`` (``IsotropicGaussian`` does).

``match_hmc(pdf, variable_name)``       whole-transition kernel for this PDF?
``match_leapfrog(pdf, variable_name)``  fused ``_leapfrog`` (``hmc.py:92-125``)?
``match_energy(pdf, variable_name)``    one-launch ``-log_prob + 0.5 sum p**2``?

``HMCSampler`` (``hmc.py:136-164``), ``spec = (kind, *params)``:

``covers(sampler, spec, D, C)``         False = use the per-step tier for this shape
``hmc(sampler, spec, q0, p0, u, accepted, adapt) -> q_out``
                                        one transition of every chain; fills ``accepted``
                                        (uint8 ``[C]``), updates ``sampler.n_accepted`` /
                                        ``_dt_chain`` / ``last_e_before`` / ``last_e_after``
``hmc_rng(sampler, spec, q0, shape) -> state or None``
                                        the same with the draws made by the kernel itself
                                        (called when no draws were supplied; None = not this
                                        time, the sampler draws and calls ``hmc``)
``hmc_n(sampler, spec, n, thin, p0, u, record, out, q0, shape) -> (handled, records)``
                                        n transitions in one launch (else ``sample_n`` loops)
``leapfrog(sampler, spec, q, p, dt, dt_chain, nsteps, mode, q_from) -> bool``
                                        integrate in place; False = not this shape
``energy(sampler, spec, q0) -> callable(x, momentum) or None``

``Likelihood`` (``likelihoods.py:141-155``), keyed by the pair of model kinds the
forward / error model advertise with ``native_spec() -> (model kind, model)``:

``likelihood={(fwd kind, err kind): (log_prob, gradient)}`` with
``log_prob(likelihood, fwm, em, fwm_vars, em_vars) -> tensor or None`` (None = evaluate
the models as written).

``GibbsSampler`` (``gibbs.py:136-151``):

``gibbs(gibbs, n, thin, record) -> (handled, records)``   n sweeps in one launch

``extras``: anything else a kind wants to expose under a name (``get(kind).extras``).

Registering a fourth model from user code::

    from binf_amd import native
    native.register('my_model', match_hmc=..., hmc=...)

and ``HMCSampler(MyPosterior(...), ...).sample()`` runs it -- see
``tests/test_gpu_registry.py``.
"""
from collections import OrderedDict

_HOOKS = ('match_hmc', 'match_leapfrog', 'match_energy', 'covers', 'hmc', 'hmc_rng', 'hmc_n',
          'leapfrog', 'energy', 'gibbs')


class Kind(object):
    """One registered kind: ``name``, the hooks above as attributes (None when not
    given), ``likelihood`` (dict) and ``extras`` (dict)."""

    def __init__(self, name, likelihood=None, extras=None, **hooks):
        unknown = set(hooks) - set(_HOOKS)
        if unknown:
            raise TypeError('unknown hook(s) for kind %r: %s' % (name, ', '.join(sorted(unknown))))
        self.name = name
        for h in _HOOKS:
            fn = hooks.get(h)
            if fn is not None and not callable(fn):
                raise TypeError('hook %s of kind %r is not callable' % (h, name))
            setattr(self, h, fn)
        self.likelihood = dict(likelihood or {})
        for pair, fns in self.likelihood.items():
            if not (isinstance(pair, tuple) and len(pair) == 2 and len(fns) == 2):
                raise TypeError('likelihood hooks of kind %r: {(forward kind, error kind): '
                                '(log_prob, gradient)}' % (name,))
        self.extras = dict(extras or {})

    def __repr__(self):
        return '<fused kind %r: %s>' % (self.name, ', '.join(
            [h for h in _HOOKS if getattr(self, h) is not None] +
            ['likelihood%s' % (list(self.likelihood),)] * bool(self.likelihood)) or 'no hooks')


_kinds = OrderedDict()


def register(kind, replace=False, **hooks):
    """Register (or, with ``replace=True``, re-register) the kind named ``kind``;
    returns the :class:`Kind`.  Recognition hooks are consulted in registration
    order."""
    if not isinstance(kind, str) or not kind:
        raise TypeError('a kind is named by a non-empty string')
    if kind in _kinds and not replace:
        raise ValueError('fused kind %r is already registered (replace=True to override)' % kind)
    k = Kind(kind, **hooks)
    _kinds[kind] = k
    return k


def unregister(kind):
    _kinds.pop(kind, None)


def get(kind):
    """The :class:`Kind` registered under ``kind`` (a name, or a spec tuple whose first
    entry is the name), or None."""
    if isinstance(kind, tuple):
        kind = kind[0] if kind else None
    return _kinds.get(kind)


def kinds():
    return list(_kinds.values())


def match(what, pdf, variable_name):
    """First registered kind whose ``match_<what>`` recognises ``pdf``:
    ``(kind name, *params)`` or None."""
    attr = 'match_' + what
    for k in _kinds.values():
        fn = getattr(k, attr)
        if fn is None:
            continue
        params = fn(pdf, variable_name)
        if params is not None:
            return (k.name,) + tuple(params)
    return None


def likelihood_hooks(forward_kind, error_kind):
    """``(log_prob, gradient)`` registered for this pair of model kinds, or None."""
    for k in _kinds.values():
        fns = k.likelihood.get((forward_kind, error_kind))
        if fns is not None:
            return fns
    return None


def model_pair(likelihood):
    """``(forward spec, error spec, hooks)`` if both models of ``likelihood``
    advertise a native kind (``native_spec()``) and some registered kind handles the
    pair, else None."""
    fs = getattr(likelihood.forward_model, 'native_spec', lambda: None)()
    es = getattr(likelihood.error_model, 'native_spec', lambda: None)()
    if fs is None or es is None:
        return None
    hooks = likelihood_hooks(fs[0], es[0])
    if hooks is None:
        return None
    return fs, es, hooks


def gibbs_sweeps(gibbs, n, thin, record):
    """Offer ``n`` sweeps of ``gibbs`` to every kind with a ``gibbs`` hook:
    ``(handled, records)``."""
    for k in _kinds.values():
        if k.gibbs is not None:
            handled, rec = k.gibbs(gibbs, n, thin, record)
            if handled:
                return True, rec
    return False, None

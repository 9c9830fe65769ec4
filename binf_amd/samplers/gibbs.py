"""
Gibbs sampling over named variables.  Mirror of reference
``binf/samplers/gibbs.py:11-190`` (without the csb base class: its
``State`` wrapper is never needed because no csb sampler exists here).

One ``sample()`` = one sweep: for every variable in ALPHABETICAL order, refresh
the conditional PDFs' parameters from the current state, let that variable's
subsampler draw, store the draw (pinned by
``binf/tests/samplers/gibbs.py:104-112``).  With chain-batched subsamplers the
sweep advances all C chains at once; the state's values are device tensors.
"""
from collections import OrderedDict


def _is_tensor(x):
    import torch
    return isinstance(x, torch.Tensor)


class GibbsSampler(object):

    # Keep the conditional PDFs' tensor-valued parameters at FIXED ADDRESSES: private copies
    # made at set-up and refreshed IN PLACE before every sub-step (one small copy each),
    # instead of handing the conditionals the state's own tensors -- so that a subsampler
    # which replays its transition from a HIP graph (``HMCSampler(graph=True)``) finds what it
    # captured.  Same values, same bits.  None = exactly when a subsampler asks for a graph.
    stable_parameters = None

    def __init__(self, pdf, state, subsamplers):
        self._state = state
        self._pdf = pdf
        self._subsamplers = subsamplers
        self._conditional_pdfs = {}
        self._setup_conditional_pdfs()
        self._update_subsampler_states()

    # -- conditional PDFs ----------------------------------------------------
    def _setup_conditional_pdfs(self):
        """For every state variable: the full PDF with all OTHER variables
        fixed to their current values; that conditional is handed to the
        variable's subsampler (reference ``:40-52``)."""
        current = self._state.variables
        already_fixed = {x: self._pdf[x].value for x in self._pdf.parameters
                         if x in self._pdf._original_variables}
        self._own = {}                    # (variable, parameter) -> the conditional's private buffer
        for var in current:
            fixed = {x: v for x, v in current.items() if x != var}
            fixed.update(already_fixed)
            cond = self._pdf.conditional_factory(**fixed)
            self._conditional_pdfs[var] = cond
            self._subsamplers[var].pdf = cond

    def _stable(self):
        if self.stable_parameters is not None:
            return bool(self.stable_parameters)
        return any(getattr(s, 'graph', False) for s in self._subsamplers.values())

    def _update_conditional_pdf_params(self):
        current = self._state.variables
        stable = self._stable()
        own = getattr(self, '_own', None)
        if own is None:
            own = self._own = {}
        for var, cond in self._conditional_pdfs.items():
            for param in cond.parameters:
                if param in current:
                    new = current[param]
                    if stable and _is_tensor(new):
                        # the conditional's OWN buffer, refreshed in place (see stable_parameters);
                        # never a tensor of the state, which callers may hold
                        buf = own.get((var, param))
                        if buf is None or buf.shape != new.shape or buf.dtype != new.dtype or \
                                buf.device != new.device:
                            buf = own[(var, param)] = new.clone()
                        elif buf is not new:
                            buf.copy_(new)
                        new = buf
                    cond[param].set(new)

    def _checkstate(self, state):
        if not type(state) == dict:
            raise TypeError(state)

    @property
    def pdf(self):
        return self._pdf

    @pdf.setter
    def pdf(self, value):
        self._pdf = value
        self._setup_conditional_pdfs()

    @property
    def state(self):
        return self._state

    @property
    def subsamplers(self):
        return self._subsamplers

    def update_samplers(self, **samplers):
        self._subsamplers.update(**samplers)

    # -- sweep -----------------------------------------------------------------
    def _update_subsampler_states(self):
        current = self._state.variables
        for variable in current:
            self._subsamplers[variable].state = current[variable]

    def _update_state(self, **variables):
        self._state.update_variables(**variables)

    # One sweep as ONE launch where a registered kind (binf_amd/native.py: ``gibbs`` hook)
    # recognises the scheme -- the example's: the multi-sweep kernel with n = 1, bit-identical
    # to the per-variable loop below; ~8 launches otherwise.  False: always the per-variable
    # loop (the tests hold the fused launch to it).
    fused_sweep = True

    def sample(self):
        self._update_subsampler_states()          # "needed for RE", :144
        if self.fused_sweep:
            from binf_amd import native
            if native.gibbs_sweeps(self, 1, 1, False)[0]:
                self._update_conditional_pdf_params()
                return self._state
        for var in sorted(list(self._pdf.variables)):
            self._update_conditional_pdf_params()
            new = self._subsamplers[var].sample()
            self._update_state(**{var: new})
        return self._state

    def sample_n(self, n, thin=1, record=True):
        """``n`` sweeps -- the ``for i in range(n): gips.sample()`` loop of the
        reference's ``example_script.py:33-34`` -- returning the recorded states
        as ``{variable: tensor [n // thin, ...]}`` (the state after sweeps
        ``thin, 2*thin, ...``; ``example_script.py:41`` keeps every 20th) or None
        if ``record`` is false.

        Where a registered kind recognises the scheme (``binf_amd.native``; the
        example's: HMC or RWMC on the polynomial coefficients + the conjugate Gamma
        draw of the precision, draws from a ``DeviceRNG`` or from the reference's
        host stream) this is ONE kernel launch with every chain's state in registers
        between the sweeps, bit-identical to n ``sample()`` calls
        (``csrc/gibbs_poly.hip``); any other scheme loops over ``sample()``."""
        n, thin = int(n), int(thin)
        if n < 1 or thin < 1:
            raise ValueError('sample_n: n >= 1 and thin >= 1 required')
        self._update_subsampler_states()
        self._update_conditional_pdf_params()
        from binf_amd import native
        handled, rec = native.gibbs_sweeps(self, n, thin, record)
        if handled:
            self._update_conditional_pdf_params()
            return rec
        import torch
        kept = {}
        for i in range(n):
            s = self.sample()
            if record and (i + 1) % thin == 0:
                for k, v in s.variables.items():
                    kept.setdefault(k, []).append(v.clone() if isinstance(v, torch.Tensor) else v)
        if not kept:
            return None
        return {k: (torch.stack(v) if isinstance(v[0], torch.Tensor) else v)
                for k, v in kept.items()}

    # -- checkpoint / resume (binf_amd/checkpoint.py) ----------------------------------
    def state_dict(self):
        subs = {}
        for var, sub in self._subsamplers.items():
            fn = getattr(sub, 'state_dict', None)
            subs[var] = fn() if fn is not None else None
        return {'variables': self._state.variables, 'subsamplers': subs}

    def load_state_dict(self, d):
        from binf_amd.checkpoint import like
        current = self._state.variables
        self._state.update_variables(**{k: like(v, current.get(k)) for k, v in d['variables'].items()})
        for var, sd in d['subsamplers'].items():
            sub = self._subsamplers[var]
            if sd is not None and hasattr(sub, 'load_state_dict'):
                sub.load_state_dict(sd)
        self._update_subsampler_states()
        self._update_conditional_pdf_params()

    # -- statistics ------------------------------------------------------------
    def _calc_pacc(self):
        """Not applicable (reference ``gibbs.py:153-157``)."""

    def _propose(self):
        """Not applicable (reference ``gibbs.py:159-163``)."""

    @property
    def last_draw_stats(self):
        # a subsampler's stats are keyed by ITS variable name, which must be
        # the Gibbs key (reference :165-174)
        return {k: v.last_draw_stats[k] for k, v in self._subsamplers.items()
                if getattr(v, 'last_draw_stats', None) is not None}

    @property
    def sampling_stats(self):
        out = OrderedDict()
        for s in self._subsamplers.values():
            if 'sampling_stats' in dir(s):
                out.update(s.sampling_stats)
        return out

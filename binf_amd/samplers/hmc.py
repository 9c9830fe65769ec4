"""
Chain-batched Hamiltonian Monte Carlo behind the reference's ``HMCSampler``
surface (``binf/samplers/hmc.py:15-191``).

One ``sample()`` call advances C independent chains by one HMC transition:
momentum draw, E_before, ``nsteps`` leapfrog steps, E_after, Metropolis
accept, optional step-size adaption.  The state is a ``[C x D]`` fp64 ROCm
tensor (a ``[D]`` tensor is one chain).  All arithmetic runs in HIP kernels:

* fused tier  -- the PDF advertises a native trajectory kernel
  (``pdf.native_hmc_spec(name) -> (kind, *params)``, a kind registered with
  ``binf_amd.native``): the whole transition is one launch with q, p held in
  registers.  The sampler never names a model: every kind registers itself
  (``binf_amd/native.py``) from the module that defines its model;
* generic tier -- any PDF with ``log_prob`` / ``gradient`` returning batched
  tensors: kick / drift / energy / accept are separate launches around the
  user's gradient.

There is no CPU path: without the HIP library this module raises.
"""
from collections import namedtuple

import torch

from binf_amd import _native, native
from binf_amd.samplers.rng import HostLegacyRNG

HMCSampleStats = namedtuple('HMCSampleStats', 'accepted stepsize')

_MODES = {'exact': _native.MODE_EXACT, 'fma': _native.MODE_FMA}


class HMCSampler(object):
    """Same constructor and attributes as the reference (``hmc.py:17-62``).

    Differences that follow from batching, all per chain:
    ``last_move_accepted`` is a ``[C]`` bool tensor, ``n_accepted`` a ``[C]``
    int64 tensor, ``acceptance_rate`` a ``[C]`` tensor; ``timestep`` becomes a
    ``[C]`` tensor once step-size adaption is switched on (chains adapt
    independently, as C separate reference samplers would).

    Extra keyword arguments (not in the reference):
      rng   object with ``normal(shape, device)`` / ``uniform(n, device)``;
            default :class:`HostLegacyRNG` (the reference's np.random stream).
      mode  'exact' (default; bit-identical to the numpy restatement) or 'fma'.
      record_energies  keep E_before / E_after of the last call in
            ``last_e_before`` / ``last_e_after``.
      graph  per-step tier only (PDFs without a whole-transition kernel): capture the
            transition's launches -- energies, every gradient call, kicks and drifts, the
            accept -- ONCE as a HIP graph and replay it (``True``: for batches up to
            ``GRAPH_MAX_ELEMENTS`` elements, where the launches themselves are the cost:
            2.7-2.9x at 64 ... 2048 chains x 64 ... 768 dims; ``'always'``; default
            ``False``).  Same bits as the eager launches.  The PDF's ``log_prob`` /
            ``gradient`` must then be pure device code: no host synchronisation, and whatever
            they read besides the variable must keep its ADDRESS between calls -- numbers
            are frozen into the graph.  Parameters reachable through ``pdf.parameters``
            (and a Posterior's / Likelihood's components) are watched: a changed number or a
            replaced tensor means a new capture; anything else needs ``reset_graph()``.

    The tensor returned by ``sample()`` IS the new state (no defensive copy --
    a copy would double the HBM traffic of the transition).  The sampler never
    writes into a tensor it has handed out; callers that want to modify a
    sample in place must clone it first.
    """

    def __init__(self, pdf, state, timestep, nsteps, timestep_adaption_limit=0,
                 adaption_uprate=1.05, adaption_downrate=0.95,
                 variable_name=None, rng=None, mode='exact',
                 record_energies=False, graph=False):
        if mode not in _MODES:
            raise ValueError("mode must be 'exact' or 'fma', not %r" % (mode,))
        if graph not in (False, True, 'always'):
            raise ValueError("graph must be False, True or 'always', not %r" % (graph,))
        self.graph = graph
        self._graphs, self._graph_warm, self._graph_captures = {}, set(), 0
        self.pdf = pdf
        self.state = state
        self._dt_chain, self._timestep = None, 0.0
        self.timestep = timestep          # a number, or a [C] tensor of per-chain step sizes
        self.nsteps = nsteps
        self.timestep_adaption_limit = timestep_adaption_limit
        self.adaption_uprate = adaption_uprate
        self.adaption_downrate = adaption_downrate
        self._variable_name = variable_name
        self.rng = rng if rng is not None else HostLegacyRNG()
        self.mode = mode
        self.record_energies = record_energies

        self._last_move_accepted = 0
        self.n_accepted = 0
        self.counter = 0
        self.last_e_before = None
        self.last_e_after = None
        self.accepted_history = None      # [n x C] flags of the last sample_n()
        self.fused_leapfrog = True        # use a PDF's fused leapfrog kernel if it has one
        self.fused_energy = True          # ... and its one-launch energy (native_energy_spec)
        # ... and a registered kind's whole-transition kernel: True (the kind decides where it is
        # the right launch), False (never: the per-step tier), or a value of the kind's own
        # (read by its hooks; the polynomial kind: 'group' / 'lane' / 'always')
        self.fused_transition = True

    # -- reference attributes ----------------------------------------------
    @property
    def timestep(self):
        return self._dt_chain if self._dt_chain is not None else self._timestep

    @timestep.setter
    def timestep(self, value):
        if isinstance(value, torch.Tensor) and value.dim() > 0:
            self._dt_chain = value
        else:
            self._timestep = float(value)
            self._dt_chain = None

    @property
    def leapfrog_steps(self):
        """Steps a trajectory takes: ``nsteps``, and 1 for ``nsteps < 1`` -- the reference's
        loop (``hmc.py:118-120``) runs ``nsteps - 1`` times and is followed by one more drift
        and half kick (``:122-123``) whatever ``nsteps`` is."""
        return max(1, int(self.nsteps))

    @property
    def acceptance_rate(self):
        if self.counter > 0:
            n = self.n_accepted
            if isinstance(n, torch.Tensor):
                n = n.to(torch.float64)
            return n / float(self.counter)
        return 0.0

    @property
    def variable_name(self):
        return 'HMC' if self._variable_name is None else self._variable_name

    @property
    def last_move_accepted(self):
        return self._last_move_accepted

    @property
    def last_draw_stats(self):
        return {self.variable_name: HMCSampleStats(self.last_move_accepted,
                                                   self.timestep)}

    def _copy_state(self, state):
        """Copies a state (reference ``hmc.py:127-134``: ``deepcopy``)."""
        return state.clone() if isinstance(state, torch.Tensor) else state

    def _adapt_timestep(self):
        """Reference ``hmc.py:183-191`` as a method: every chain's step size times
        ``adaption_uprate`` if its last move was accepted, else times
        ``adaption_downrate`` (quirk Q3).  ``sample()`` / ``sample_n()`` do this
        inside their kernels while ``counter < timestep_adaption_limit``; the
        method is here for callers that drive the adaption themselves."""
        acc = self._last_move_accepted
        if not isinstance(acc, torch.Tensor):
            raise ValueError('_adapt_timestep: no move has been made yet')
        if self._dt_chain is None:
            self._dt_chain = torch.full(acc.shape, float(self._timestep), dtype=torch.float64,
                                        device=acc.device)
        self._dt_chain = torch.where(acc, self._dt_chain * self.adaption_uprate,
                                     self._dt_chain * self.adaption_downrate)

    # -- one transition ------------------------------------------------------
    def sample(self, p0=None, u=None):
        """Draw one sample per chain.  ``p0`` (``[C x D]``) and ``u`` (``[C]``)
        override the random draws (parity tests, pre-generated pools)."""
        name = self._variable_name
        if not isinstance(name, str):
            # reference: pdf.log_prob(**{None: x}) -> TypeError (quirk Q1)
            raise TypeError('HMCSampler needs variable_name to sample()')
        state = _device_state(self.state)
        shape = state.shape                       # quirk Q2: arrays only
        q0 = (state if state.dim() == 2 else state.reshape(1, -1)).contiguous()
        C, D = q0.shape
        dev = q0.device
        spec = self._fused_spec(name, D, C)          # once per call: walks the posterior
        kind = native.get(spec) if spec is not None else None
        if kind is not None and p0 is None and u is None and kind.hmc_rng is not None:
            # a kernel that makes the transition's draws itself
            drawn = kind.hmc_rng(self, spec, q0, shape)
            if drawn is not None:
                return drawn
        both = getattr(self.rng, 'normal_uniform', None)
        graphed = (kind is None or kind.hmc is None) and self.graph and q0.is_cuda and \
            (self.graph == 'always' or q0.numel() <= GRAPH_MAX_ELEMENTS)
        if graphed and p0 is None and u is None and hasattr(self.rng, 'fill_normal_uniform'):
            own_p = True                           # drawn straight into the graph's input buffers
        elif p0 is None and u is None and both is not None:
            p0, u = both((C, D), C, dev)           # one launch for the transition's two draws
            own_p = True
        elif p0 is None:
            p0 = self.rng.normal((C, D), dev)
            own_p = True
        else:
            p0 = (p0 if p0.dim() == 2 else p0.reshape(1, -1)).contiguous()
            own_p = False
        if u is None and p0 is not None:
            u = self.rng.uniform(C, dev)

        adapt = (self.counter + 1) < self.timestep_adaption_limit
        if adapt and self._dt_chain is None:
            self._dt_chain = torch.full((C,), float(self._timestep),
                                        dtype=torch.float64, device=dev)
        if not isinstance(self.n_accepted, torch.Tensor):
            self.n_accepted = torch.zeros(C, dtype=torch.int64, device=dev)

        accepted = torch.empty(C, dtype=torch.uint8, device=dev)
        if kind is not None and kind.hmc is not None:
            q_out = kind.hmc(self, spec, q0, p0, u, accepted, adapt)
        elif graphed:
            q_out = self._sample_graphed(name, state, q0, p0, own_p, u, accepted, adapt)
        else:
            q_out = self._sample_generic(name, state, q0, p0, own_p, u,
                                         accepted, adapt)

        self._last_move_accepted = accepted.view(torch.bool)
        self.counter += 1
        q_out = q_out.view(shape)
        self.state = q_out
        return q_out

    # -- n transitions in one launch -------------------------------------------
    def sample_n(self, n, thin=1, p0=None, u=None, record=True, out=None):
        """``n`` consecutive ``sample()`` calls (the ``for i in range(n)`` loop
        of the reference's ``example_script.py:33-34``), returning the recorded
        states ``[n // thin, C, D]`` -- the state after transitions ``thin,
        2*thin, ...`` -- or None if ``record`` is false.  Results are
        bit-identical to calling ``sample()`` n times -- with supplied draws, and
        with a generator that draws inside the kernel too (transition i of the
        launch takes the stream position the i-th single call would take).

        For PDFs with a fused kernel this is ONE launch of the persistent
        kernel (state kept in registers between transitions); otherwise it
        loops over ``sample()``.  ``p0`` is ``[n, C, D]``, ``u`` is ``[n, C]``.
        ``out`` (``[n // thin, C, D]`` fp64, contiguous, same device) is a
        caller-owned record buffer to fill instead of a fresh allocation
        (a sample store slice; the C ABI never allocates).
        """
        name = self._variable_name
        if not isinstance(name, str):
            raise TypeError('HMCSampler needs variable_name to sample()')
        n, thin = int(n), int(thin)
        if n < 1 or thin < 1:
            raise ValueError('sample_n: n >= 1 and thin >= 1 required')
        state = _device_state(self.state)
        shape = state.shape
        q0 = (state if state.dim() == 2 else state.reshape(1, -1)).contiguous()
        C, D = q0.shape
        dev = q0.device
        nrec = n // thin
        if p0 is not None:
            p0 = p0.reshape(n, C, D)
        if u is not None:
            u = u.reshape(n, C)
        if out is not None:
            if not record or nrec < 1:
                raise ValueError('sample_n: out= given but nothing is recorded')
            if out.dtype != torch.float64 or out.device != dev or \
                    not out.is_contiguous() or out.numel() != nrec * C * D:
                raise ValueError('sample_n: out must be a contiguous fp64 [%d, %d, %d] '
                                 'tensor on %s' % (nrec, C, D, dev))
        spec = self._fused_spec(name, D)
        kind = native.get(spec) if spec is not None else None
        if kind is not None and kind.hmc_n is not None:
            handled, res = kind.hmc_n(self, spec, n, thin, p0, u, record, out, q0, shape)
            if handled:
                q_out, samples = res
                self.state = q_out.view(shape)
                if samples is None:
                    return self._no_records(shape, dev) if record else None
                return samples if state.dim() == 2 else samples.reshape((nrec,) + tuple(shape))
        # no multi-transition kernel for this PDF / shape: n single calls (each draws for
        # itself when no draws were supplied)
        rec, flags, ebs, eas = [], [], [], []
        for i in range(n):
            x = self.sample(p0=None if p0 is None else p0[i], u=None if u is None else u[i])
            if record and (i + 1) % thin == 0:
                rec.append(x)
            flags.append(self._last_move_accepted)
            ebs.append(self.last_e_before)
            eas.append(self.last_e_after)
        self.accepted_history = torch.stack(flags)
        if all(e is not None for e in ebs):
            self.last_e_before, self.last_e_after = torch.stack(ebs), torch.stack(eas)
        if not record:
            return None
        if not rec:
            return self._no_records(shape, dev)
        if out is not None:
            torch.stack(rec, out=out.view((nrec,) + tuple(rec[0].shape)))
            return out
        return torch.stack(rec)

    @staticmethod
    def _no_records(shape, dev):
        """``sample_n(n, thin)`` with ``thin > n`` records nothing: ``[0, *state shape]``."""
        return torch.empty((0,) + tuple(shape), dtype=torch.float64, device=dev)

    def _fused_spec(self, name, D, C=None):
        """The PDF's fused-kernel descriptor ``(kind, *params)`` if a registered kind
        covers this shape (and, for ``C`` chains, if its kernel is the faster
        choice -- the kind's ``covers`` hook decides), else None (generic per-step
        tier)."""
        get_spec = getattr(self.pdf, 'native_hmc_spec', None)
        spec = get_spec(name) if (get_spec is not None and self.fused_transition is not False) else None
        if spec is None:
            return None
        kind = native.get(spec)
        if kind is None or (kind.hmc is None and kind.hmc_n is None and kind.hmc_rng is None):
            return None
        if kind.covers is not None and not kind.covers(self, spec, D, C):
            return None
        return spec

    # -- generic tier --------------------------------------------------------
    def _leapfrog(self, q, p, timestep, nsteps, q_from=None):
        """Velocity-Verlet integration, IN PLACE on ``q`` and ``p`` (reference
        ``hmc.py:92-125``: half kick, ``nsteps - 1`` x [drift, kick], drift,
        half kick; ``nsteps + 1`` gradient calls).  ``timestep`` is a float or
        a ``[C]`` tensor of per-chain step sizes.  Returns ``(q, p)``.
        ``q_from`` (not in the reference): the start positions, when ``q`` is only to
        receive the end positions -- ``sample()`` then needs no copy of its state
        (``hmc.py:140-141``) where a fused kernel can read the start from elsewhere."""
        name = self._variable_name
        pdf = self.pdf
        mode = _MODES[self.mode]
        shape = q.shape
        q2, p2 = _as2d(q), _as2d(p)
        if not (q2.is_contiguous() and p2.is_contiguous()):
            raise ValueError('_leapfrog integrates in place: q and p must be contiguous')
        if isinstance(timestep, torch.Tensor):
            dt, dtc = 0.0, timestep.reshape(-1).contiguous()
        else:
            dt, dtc = float(timestep), None
        def grad(x):
            g = pdf.gradient(**{name: x.view(shape)})
            if not isinstance(g, torch.Tensor) or g.numel() != x.numel():
                # e.g. a posterior none of whose components is differentiable in `name`:
                # the reference's zero vector has one entry per DIFFERENTIABLE element
                # (posteriors.py:177-180) -- none -- and its `p -= ... * gradient(q)`
                # (hmc.py:116) fails to broadcast; the same error here
                raise ValueError('pdf.gradient(%s=...) returned %s for a state of shape %s '
                                 '(is the variable differentiable in any component?)'
                                 % (name, tuple(getattr(g, 'shape', ())) or type(g).__name__,
                                    tuple(shape)))
            return _as2d(g).contiguous()

        leap = getattr(pdf, 'native_leapfrog_spec', None)
        leap = leap(name) if (leap is not None and self.fused_leapfrog) else None
        kind = native.get(leap) if leap is not None else None
        if kind is not None and kind.leapfrog is not None:
            # the whole integration in the kind's own launches (bit-identical to the loop)
            if kind.leapfrog(self, leap, q2, p2, dt, dtc, max(1, int(nsteps)), mode,
                             None if q_from is None else _as2d(q_from)):
                return q, p
        if q_from is not None:
            q2.copy_(_as2d(q_from))
        # half kick, drift, (nsteps - 1) x [gradient, kick + next drift], half kick:
        # the reference's sequence (hmc.py:116-123) with every interior kick
        # and the drift that follows it in one pass over memory
        _native.leapfrog_kick(p2, grad(q2), dt, dtc, half=True, mode=mode)
        _native.leapfrog_drift(q2, p2, dt, dtc, mode=mode)
        for _ in range(nsteps - 1):
            _native.leapfrog_kick_drift(q2, p2, grad(q2), dt, dtc, mode=mode)
        _native.leapfrog_kick(p2, grad(q2), dt, dtc, half=True, mode=mode)
        return q, p

    def _sample_generic(self, name, state, q0, p0, own_p, u, accepted, adapt):
        pdf = self.pdf
        mode = _MODES[self.mode]
        shape = state.shape
        dtc = self._dt_chain
        # E = -log_prob + 0.5 sum p^2 (hmc.py:143,148,150): the kinetic row sum with the
        # subtraction as its epilogue (one launch; the same bits as negate, sum, add)
        E = lambda x, mom: _native.hmc_energy(
            mom, _as_chain_vector(pdf.log_prob(**{name: x.view(shape)})).contiguous())
        espec = getattr(pdf, 'native_energy_spec', None)
        espec = espec(name) if (espec is not None and self.fused_energy and q0.is_cuda) else None
        ekind = native.get(espec) if espec is not None else None
        if ekind is not None and ekind.energy is not None:
            # the PDF's terms, their sum and the kinetic energy in the kind's one launch
            # (bit-identical to the calls above); None = not this shape
            fused = ekind.energy(self, espec, q0)
            if fused is not None:
                E = fused

        p = p0 if own_p else p0.clone()
        e_before = E(q0, p)
        step = self._timestep if dtc is None else dtc
        if type(self)._leapfrog is HMCSampler._leapfrog:
            # the trajectory starts from q0 and ends in q: no copy of the state first where the
            # fused leapfrog reads its start from q0 itself
            q = torch.empty_like(q0)
            self._leapfrog(q.view(shape), p.view(shape), step, self.nsteps, q_from=q0.view(shape))
        else:
            # a subclass's integrator has the reference's signature (hmc.py:92): in place on a copy
            q = q0.clone()
            self._leapfrog(q.view(shape), p.view(shape), step, self.nsteps)
        e_after = E(q, p)
        _native.accept_select(q, q0, e_before, e_after, u, q, accepted,
                              self.n_accepted, dtc, adapt,
                              self.adaption_uprate, self.adaption_downrate)
        self.last_e_before, self.last_e_after = e_before, e_after
        return q


    # -- checkpoint / resume (binf_amd/checkpoint.py) ----------------------------------
    def state_dict(self):
        """What a fresh sampler built with the same arguments needs to continue this chain
        bit for bit: state, step sizes, counters, the generator's position."""
        rng = getattr(self.rng, 'state_dict', None)
        return {'state': self.state, 'timestep': float(self._timestep), 'dt_chain': self._dt_chain,
                'n_accepted': self.n_accepted, 'counter': int(self.counter),
                'last_move_accepted': self._last_move_accepted,
                'rng': rng() if rng is not None else None}

    def load_state_dict(self, d):
        from binf_amd.checkpoint import like
        ref = self.state
        self.state = like(d['state'], ref)
        self._timestep = float(d['timestep'])
        self._dt_chain = None if d['dt_chain'] is None else like(d['dt_chain'], ref)
        n = d['n_accepted']
        self.n_accepted = n.to(ref.device) if isinstance(n, torch.Tensor) else n
        self.counter = int(d['counter'])
        a = d['last_move_accepted']
        self._last_move_accepted = a.to(ref.device) if isinstance(a, torch.Tensor) else a
        if d.get('rng') is not None and hasattr(self.rng, 'load_state_dict'):
            self.rng.load_state_dict(d['rng'])
        self.reset_graph()

    # -- the per-step tier as one HIP graph --------------------------------------------
    def reset_graph(self):
        """Forget the captured graphs (after changing something the PDF reads that is not
        one of its registered parameters)."""
        self._graphs.clear()
        self._graph_warm.clear()

    def _graph_key(self, state, q0, adapt):
        dtc = self._dt_chain
        return (tuple(state.shape), q0.device.index, bool(adapt), float(self._timestep),
                None if dtc is None else dtc.data_ptr(), int(self.nsteps), self.mode,
                float(self.adaption_uprate), float(self.adaption_downrate),
                self.n_accepted.data_ptr(), bool(self.fused_leapfrog), bool(self.fused_energy),
                type(self)._leapfrog, _graph_signature(self.pdf))

    def _sample_graphed(self, name, state, q0, p0, own_p, u, accepted, adapt):
        """``_sample_generic`` replayed from a HIP graph: first call with a configuration
        eager (it also warms every lazily initialised piece up), second call captured, every
        later one replayed.  Inputs are copied into the graph's own buffers and the new state
        out of them, so no tensor handed out is ever written again."""
        key = self._graph_key(state, q0, adapt)
        entry = self._graphs.get(key)
        if entry is None and key in self._graph_warm:
            entry = self._capture(key, name, state, q0, adapt)
        if entry is None:
            # first call with this configuration (it also warms everything up), or graphs are off
            if self.graph and key not in self._graph_warm:
                self._graph_warm.add(key)
                if len(self._graph_warm) > 4 * GRAPH_MAX_CAPTURES:
                    self._graph_gives_up('%d configurations seen, none twice' % len(self._graph_warm))
            if p0 is None:                             # the draws sample() left to the graph's buffers
                both = getattr(self.rng, 'normal_uniform', None)
                p0, u = both(tuple(q0.shape), q0.shape[0], q0.device) if both is not None else \
                    (self.rng.normal(tuple(q0.shape), q0.device), self.rng.uniform(q0.shape[0], q0.device))
            return self._sample_generic(name, state, q0, p0, own_p, u, accepted, adapt)
        g, q_in, p_in, u_in, acc, q_out, e_b, e_a = entry
        q_in.copy_(q0)
        if p0 is None:
            self.rng.fill_normal_uniform(p_in, u_in)   # same values and stream positions as eager
        else:
            p_in.copy_(p0)
            u_in.copy_(u.reshape(u_in.shape))
        g.replay()
        accepted.copy_(acc)
        # (record_energies: private copies; otherwise the graph's own buffers, valid until the next call)
        self.last_e_before, self.last_e_after = (e_b.clone(), e_a.clone()) if self.record_energies \
            else (e_b, e_a)
        return q_out.clone()

    def _graph_gives_up(self, why):
        """Every call a new configuration (e.g. a Gibbs loop that REPLACES parameter tensors
        instead of updating them in place), or a PDF that cannot be captured: eager launches
        from now on, said once."""
        import warnings
        warnings.warn('HMCSampler(graph=%r): %s; graph mode switched off for this sampler'
                      % (self.graph, why))
        self.graph = False
        self.reset_graph()

    def _capture(self, key, name, state, q0, adapt):
        if self._graph_captures >= 4 * GRAPH_MAX_CAPTURES:
            self._graph_gives_up('%d captures' % self._graph_captures)
            return None
        q_in, p_in = torch.empty_like(q0), torch.empty_like(q0)
        u_in = torch.empty(q0.shape[0], dtype=torch.float64, device=q0.device)
        acc = torch.empty(q0.shape[0], dtype=torch.uint8, device=q0.device)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                q_out = self._sample_generic(name, q_in.view(state.shape), q_in, p_in, True, u_in,
                                             acc, adapt)
                e_b, e_a = self.last_e_before, self.last_e_after
        except Exception as e:                          # noqa: BLE001 -- whatever the PDF did
            torch.cuda.synchronize()
            self._graph_gives_up('the transition could not be captured (%s: %s) -- does the PDF '
                                 'synchronise with the host?' % (type(e).__name__, str(e).split('\n')[0][:200]))
            return None
        self._graph_captures += 1
        if len(self._graphs) >= GRAPH_MAX_CAPTURES:
            self._graphs.pop(next(iter(self._graphs)))
        entry = (g, q_in, p_in, u_in, acc, q_out, e_b, e_a)
        self._graphs[key] = entry
        return entry


# graph=True captures batches up to this many elements (beyond it the kernels, not their
# launches, are the cost: 4096 x 1024 measured 0.98x), and keeps at most this many graphs
GRAPH_MAX_ELEMENTS = 1 << 21
GRAPH_MAX_CAPTURES = 8


def _graph_signature(pdf, depth=0):
    """What a captured transition froze of ``pdf``: the numbers and the tensor ADDRESSES of
    its registered parameters, through a Posterior's components and a Likelihood's models."""
    sig = []
    names = getattr(pdf, 'parameters', None)
    if names is not None and depth < 4:
        try:
            for n in names:
                v = pdf[n].value
                if isinstance(v, torch.Tensor):
                    sig.append((n, v.data_ptr(), tuple(v.shape)))
                elif isinstance(v, (int, float)):
                    sig.append((n, float(v)))
                else:
                    sig.append((n, id(v)))
        except Exception:                               # noqa: BLE001 -- a duck-typed pdf
            sig.append(('?', id(pdf)))
    for attr in ('_components', '_likelihoods', '_priors'):
        comps = getattr(pdf, attr, None)
        if isinstance(comps, dict):
            for n in sorted(comps):
                sig.append((attr, n, _graph_signature(comps[n], depth + 1)))
            break
    for attr in ('forward_model', 'error_model'):
        m = getattr(pdf, attr, None)
        if m is not None and depth < 4:
            sig.append((attr, _graph_signature(m, depth + 1)))
    return (id(pdf), tuple(sig))


def _device_state(state):
    """The sampler state as it is -- except a numpy array, refused by name (the reference's
    states are numpy arrays; here they are ROCm tensors and there is no CPU path).  Anything
    without a ``shape`` keeps failing the reference's way (quirk Q2: AttributeError)."""
    if type(state).__module__ == 'numpy':
        raise TypeError('the sampler state is a numpy %s; binf_amd keeps states as ROCm (cuda) fp64 '
                        'tensors and evaluates on the GPU only (no CPU path)' % type(state).__name__)
    return state


def _fill(rng, kind, out):
    """One draw of ``rng`` into the contiguous buffer ``out`` (generators without
    fill_* methods: draw, then copy)."""
    fill = getattr(rng, 'fill_' + kind, None)
    if fill is not None:
        fill(out)
    elif kind == 'normal':
        out.copy_(rng.normal(tuple(out.shape), out.device))
    else:
        out.copy_(rng.uniform(out.numel(), out.device).reshape(out.shape))


def _as2d(x):
    return x if x.dim() == 2 else x.reshape(1, -1)


def _as_chain_vector(x):
    if not isinstance(x, torch.Tensor):
        raise TypeError('pdf.log_prob must return a tensor with one value per '
                        'chain, got %r' % type(x))
    return x.reshape(-1)

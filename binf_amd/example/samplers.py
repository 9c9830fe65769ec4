"""
Subsamplers of the example application (mirror of reference
``binf/example/samplers.py``), chain-batched: the conjugate Gamma update of
the noise precision, a random-walk Metropolis sampler, and the Gibbs factory.
"""
from collections import namedtuple

import numpy as np
import torch

from binf_amd import _native

RWMCSampleStats = namedtuple('RWMCSampleStats', 'acceptance_rate')


class GammaSampler(object):
    """Conjugate draw of the precision: ``tau = Gamma(shape) / rate`` with
    ``shape = 0.5*n + prior.shape - 1`` (as written in the reference -- one
    less than the textbook value) and ``rate = -L.log_prob(coefficients,
    precision=1) + prior.rate`` (reference ``:27-51``).  One draw per chain.

    The prior is read from ``self.pdf`` -- under a GibbsSampler that is the
    CONDITIONAL posterior, whose GammaPrior copy has rate == shape (quirk Q6).

    Where the Gamma(shape, 1) variates come from (reference ``:47``:
    ``np.random.gamma(shape)``):

    * ``rng`` -- a generator with a ``gamma(shape, n_chains, device)`` method
      (:class:`binf_amd.samplers.rng.DeviceRNG`): drawn ON THE DEVICE, one
      per chain, keyed by the global chain index; nothing crosses PCIe;
    * ``gamma`` -- a callable ``(shape, n_chains, device) -> [C]`` tensor;
    * neither: ``np.random.gamma`` from the global legacy stream like the
      reference, uploaded (parity mode; a host draw + H2D copy per sweep).
    """

    def __init__(self, pdf, state, gamma=None, rng=None):
        self.pdf = pdf
        self.state = state
        if gamma is None and rng is not None:
            gamma = getattr(rng, 'gamma', None)
            if gamma is None:
                raise TypeError('GammaSampler: rng has no gamma(shape, n, device) method')
        self.gamma = gamma

    # -- checkpoint / resume (binf_amd/checkpoint.py) ----------------------------------
    def state_dict(self):
        rng = getattr(self.gamma, '__self__', None)            # the generator whose gamma() draws
        fn = getattr(rng, 'state_dict', None)
        return {'state': self.state, 'rng': fn() if fn is not None else None}

    def load_state_dict(self, d):
        from binf_amd.checkpoint import like
        self.state = like(d['state'], self.state)
        rng = getattr(self.gamma, '__self__', None)
        if d.get('rng') is not None and hasattr(rng, 'load_state_dict'):
            rng.load_state_dict(d['rng'])

    def _get_prior(self):
        from binf_amd.example.priors import GammaPrior
        prior = [p for p in self.pdf.priors.values()
                 if 'precision' in p.variables][0]
        if not isinstance(prior, GammaPrior):
            raise NotImplementedError('Prior for precision is not a Gamma '
                                      'distribution')
        return prior

    def _calculate_shape(self):
        prior = self._get_prior()
        n_data_points = len(self.pdf.likelihoods['points'].error_model.ys)
        return 0.5 * n_data_points + prior.shape - 1

    def _unit_precision_log_prob(self):
        args = dict(coefficients=self.pdf['coefficients'].value, precision=1.0)
        return self.pdf.likelihoods['points'].log_prob(**args)

    def _calculate_rate(self):
        return -self._unit_precision_log_prob() + self._get_prior().rate

    def sample(self, state=42):
        shape = self._calculate_shape()
        lp1 = _native.require_device(self._unit_precision_log_prob(),
                                     'the likelihood log-prob')
        C = lp1.numel()
        if self.gamma is not None:
            g = self.gamma(shape, C, lp1.device)
        else:
            g = torch.from_numpy(np.random.gamma(shape, size=C)).to(lp1.device)
        self.state = _native.gamma_precision_update(
            g, lp1.reshape(-1).contiguous(), self._get_prior().rate)
        return self.state


class RWMCSampler(object):
    """Random-walk Metropolis on ``coefficients`` (reference ``:54-92``):
    uniform proposal of half-width ``stepsize``, accept with
    ``random() < np.exp(-(E_new - E_old))``.  Per-chain acceptance counts.

    ``rng=None``: the reference's draws -- ``np.random.uniform(size=K)`` then
    ``np.random.random()`` from the global legacy stream (for C chains:
    ``size=(C, K)``, ``size=C``), uploaded; one chain consumes the stream
    exactly as the reference does.  ``rng=DeviceRNG(...)``: both draws are
    generated inside the proposal / accept kernels from the generator's Philox
    stream, keyed by the global chain index -- no host draw, no H2D copy, and
    a shard of a run draws what the whole run draws for its chains.
    """

    def __init__(self, pdf, state, stepsize, rng=None):
        self.pdf = pdf
        self.state = state
        self.stepsize = stepsize
        self.rng = rng
        self._n_moves = 0
        self._n_accepted_moves = 0
        self.last_move_accepted = None

    @property
    def last_draw_stats(self):
        return {'coefficients': RWMCSampleStats(self.acceptance_rate)}

    # -- checkpoint / resume (binf_amd/checkpoint.py; a run on the host stream also saves a
    # ``HostLegacyRNG()``, which IS that stream) ------------------------------------------
    def state_dict(self):
        fn = getattr(self.rng, 'state_dict', None)
        return {'state': self.state, 'stepsize': float(self.stepsize), 'n_moves': int(self._n_moves),
                'n_accepted_moves': self._n_accepted_moves, 'last_move_accepted': self.last_move_accepted,
                'rng': fn() if fn is not None else None}

    def load_state_dict(self, d):
        from binf_amd.checkpoint import like
        ref = self.state
        self.state = like(d['state'], ref)
        self.stepsize, self._n_moves = float(d['stepsize']), int(d['n_moves'])
        n, a = d['n_accepted_moves'], d['last_move_accepted']
        self._n_accepted_moves = n.to(ref.device) if isinstance(n, torch.Tensor) else n
        self.last_move_accepted = a.to(ref.device) if isinstance(a, torch.Tensor) else a
        if d.get('rng') is not None and hasattr(self.rng, 'load_state_dict'):
            self.rng.load_state_dict(d['rng'])

    @property
    def acceptance_rate(self):
        if self._n_moves > 0:
            n = self._n_accepted_moves
            if isinstance(n, torch.Tensor):
                n = n.to(torch.float64)
            return n / float(self._n_moves)
        return 0.0

    def sample(self, change=None, u=None):
        """One Metropolis move per chain.  ``change`` (``[C x K]``) and ``u``
        (``[C]``) override the draws (tests, pre-generated pools)."""
        state = _native.require_device(self.state, 'the RWMC state')
        s2 = (state if state.dim() == 2 else state.reshape(1, -1)).contiguous()
        C, K = s2.shape
        dev = s2.device
        lp_old = self.pdf.log_prob(coefficients=state)               # E_old = -lp_old
        seed = off_c = off_u = coff = 0
        if self.rng is not None and hasattr(self.rng, 'next_offset'):
            # device draws: one stream position for the proposal, one for the test
            seed, coff = self.rng.seed, int(getattr(self.rng, 'chain_offset', 0))
            off_c = self.rng.next_offset() if change is None else 0
            off_u = self.rng.next_offset() if u is None else 0
        else:
            if change is None:
                shape = (K,) if state.dim() == 1 else (C, K)
                change = torch.from_numpy(np.random.uniform(
                    low=-self.stepsize, high=self.stepsize, size=shape)).to(dev)
        if change is not None:
            change = change.reshape(C, K).contiguous()
        proposal = _native.rwmc_propose(s2, self.stepsize, change, seed, off_c, coff)
        lp_new = self.pdf.log_prob(coefficients=proposal.view(state.shape))
        if u is None and not (self.rng is not None and hasattr(self.rng, 'next_offset')):
            u = torch.from_numpy(np.random.random(size=C)).to(dev)
        accepted = torch.empty(C, dtype=torch.uint8, device=dev)
        if not isinstance(self._n_accepted_moves, torch.Tensor):
            self._n_accepted_moves = torch.zeros(C, dtype=torch.int64, device=dev)
        _native.rwmc_accept(proposal, s2, lp_old.reshape(-1).contiguous(),
                            lp_new.reshape(-1).contiguous(), proposal, accepted,
                            self._n_accepted_moves, u, seed, off_u, coff)
        self.last_move_accepted = accepted.view(torch.bool)
        self.state = proposal.view(state.shape)
        self._n_moves += 1
        return self.state


def make_sampler(posterior, rwmc_stepsize, start_state, rng=None):
    """The reference's factory (``:94-111``): RWMC for the coefficients,
    conjugate Gamma for the precision, wrapped in a GibbsSampler.  ``rng``
    (not in the reference): a :class:`DeviceRNG` makes every draw of the sweep
    a device draw (nothing crosses PCIe); default = the reference's global
    ``np.random`` stream."""
    from binf_amd.samplers.gibbs import GibbsSampler
    coeffs = start_state.variables['coefficients']
    precision = start_state.variables['precision']
    coefficients_sampler = RWMCSampler(
        posterior.conditional_factory(precision=precision), coeffs,
        rwmc_stepsize, rng=rng)
    precision_sampler = GammaSampler(
        posterior.conditional_factory(coefficients=coeffs), precision, rng=rng)
    return GibbsSampler(posterior, start_state,
                        {'coefficients': coefficients_sampler,
                         'precision': precision_sampler})


def make_hmc_sampler(posterior, timestep, nsteps, start_state, **hmc_kwargs):
    """The build's counterpart for BASELINE configs C1/C4: HMC for the
    coefficients inside the same Gibbs scheme (this wiring does not exist in
    the reference, which never runs its HMCSampler)."""
    from binf_amd.samplers.gibbs import GibbsSampler
    from binf_amd.samplers.hmc import HMCSampler
    coeffs = start_state.variables['coefficients']
    precision = start_state.variables['precision']
    gamma = hmc_kwargs.pop('gamma', None)
    coefficients_sampler = HMCSampler(
        posterior.conditional_factory(precision=precision), coeffs, timestep,
        nsteps, variable_name='coefficients', **hmc_kwargs)
    # a device generator given for the HMC draws also serves the Gamma draw
    # (GammaSampler: device gamma when an rng with .gamma is there)
    rng = hmc_kwargs.get('rng')
    precision_sampler = GammaSampler(
        posterior.conditional_factory(coefficients=coeffs), precision,
        gamma=gamma, rng=rng if (gamma is None and hasattr(rng, 'gamma')) else None)
    return GibbsSampler(posterior, start_state,
                        {'coefficients': coefficients_sampler,
                         'precision': precision_sampler})

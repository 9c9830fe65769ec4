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

    ``gamma`` (optional): callable ``(shape, n_chains, device) -> [C]`` tensor
    of Gamma(shape, 1) variates; default draws ``np.random.gamma`` from the
    global legacy stream like the reference (``:47``).
    """

    def __init__(self, pdf, state, gamma=None):
        self.pdf = pdf
        self.state = state
        self.gamma = gamma

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
    ``random() < exp(-(E_new - E_old))``.  Per-chain acceptance counts."""

    def __init__(self, pdf, state, stepsize):
        self.pdf = pdf
        self.state = state
        self.stepsize = stepsize
        self._n_moves = 0
        self._n_accepted_moves = 0

    @property
    def last_draw_stats(self):
        return {'coefficients': RWMCSampleStats(self.acceptance_rate)}

    @property
    def acceptance_rate(self):
        if self._n_moves > 0:
            n = self._n_accepted_moves
            if isinstance(n, torch.Tensor):
                n = n.to(torch.float64)
            return n / float(self._n_moves)
        return 0.0

    def sample(self):
        state = _native.require_device(self.state, 'the RWMC state')
        E_old = -self.pdf.log_prob(coefficients=state)
        s2 = state if state.dim() == 2 else state.reshape(1, -1)
        C, K = s2.shape
        dev = s2.device
        shape = (K,) if state.dim() == 1 else (C, K)
        change = torch.from_numpy(np.random.uniform(
            low=-self.stepsize, high=self.stepsize, size=shape)).to(dev)
        proposal = state + change
        E_new = -self.pdf.log_prob(coefficients=proposal)
        u = torch.from_numpy(np.random.random(size=C)).to(dev)
        accepted = torch.empty(C, dtype=torch.uint8, device=dev)
        if not isinstance(self._n_accepted_moves, torch.Tensor):
            self._n_accepted_moves = torch.zeros(C, dtype=torch.int64, device=dev)
        p2 = proposal if proposal.dim() == 2 else proposal.reshape(1, -1)
        _native.accept_select(p2.contiguous(), s2.contiguous(),
                              E_old.reshape(-1), E_new.reshape(-1), u, p2,
                              accepted, self._n_accepted_moves)
        self.state = p2.view(state.shape)
        self._n_moves += 1
        return self.state


def make_sampler(posterior, rwmc_stepsize, start_state):
    """The reference's factory (``:94-111``): RWMC for the coefficients,
    conjugate Gamma for the precision, wrapped in a GibbsSampler."""
    from binf_amd.samplers.gibbs import GibbsSampler
    coeffs = start_state.variables['coefficients']
    precision = start_state.variables['precision']
    coefficients_sampler = RWMCSampler(
        posterior.conditional_factory(precision=precision), coeffs,
        rwmc_stepsize)
    precision_sampler = GammaSampler(
        posterior.conditional_factory(coefficients=coeffs), precision)
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
    precision_sampler = GammaSampler(
        posterior.conditional_factory(coefficients=coeffs), precision,
        gamma=gamma)
    return GibbsSampler(posterior, start_state,
                        {'coefficients': coefficients_sampler,
                         'precision': precision_sampler})

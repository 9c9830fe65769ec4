"""
Pairwise-distance-restraint model (BASELINE config C5).  BUILD-DEFINED: the
reference only names the chromatin application (``README.rst:9``) and holds no
code for it, so this follows the shape of the reference's plug-in surface
(``AbstractForwardModel`` / ``AbstractErrorModel``) and its Gaussian error
model (``binf/example/likelihood.py:40-68``); parity is against the numpy
formulation kept with the test infrastructure only.

``coordinates`` is ``[C x 3n]`` (bead-major: x, y, z of bead 0, then bead 1 ...);
the mock data are the n(n-1)/2 pair distances in ``numpy.triu_indices(n, 1)``
order.
"""
import numpy as np
import torch

from binf_amd import ArrayParameter, _native, native
from binf_amd.example.likelihood import GaussianErrorModel
from binf_amd.model.forwardmodels import AbstractForwardModel


KIND = 'pairdist'        # the name this module's hooks are registered under


class DistanceForwardModel(AbstractForwardModel):

    def __init__(self, n_beads):
        super(DistanceForwardModel, self).__init__('pair_distances')
        self.n_beads = int(n_beads)
        self._pairs = np.triu_indices(self.n_beads, 1)
        self._dev = {}
        self._register_variable('coordinates', differentiable=True)
        self.update_var_param_types(coordinates=ArrayParameter)
        self._set_original_variables()

    @property
    def n_pairs(self):
        return len(self._pairs[0])

    def pair_index(self, device):
        if device not in self._dev:
            I, J = self._pairs
            self._dev[device] = (
                torch.from_numpy(I.astype(np.int32)).to(device),
                torch.from_numpy(J.astype(np.int32)).to(device))
        return self._dev[device]

    def _evaluate(self, coordinates):
        _native.require_device(coordinates, 'coordinates')
        x = coordinates if coordinates.dim() == 2 else coordinates.reshape(1, -1)
        I, J = self.pair_index(x.device)
        out = _native.pairdist_forward(x.contiguous(), I, J)
        return out if coordinates.dim() == 2 else out.reshape(-1)

    def _evaluate_jacobi_matrix(self, coordinates):
        raise NotImplementedError(
            'the [3n x n(n-1)/2] Jacobian is never formed; the Likelihood uses '
            'the fused all-pairs gradient kernel')

    def clone(self):
        copy = self.__class__(self.n_beads)
        copy._dev = self._dev
        self._set_parameters(copy)
        return copy

    def native_spec(self):
        from binf_amd.example.likelihood import _unchanged
        if _unchanged(self, DistanceForwardModel, ('_evaluate', '_evaluate_jacobi_matrix')):
            return ('pairdist', self)
        return None


class DistanceErrorModel(GaussianErrorModel):
    """Gaussian error model on the pair distances; ``ys`` are the target
    distances in pair order.  Keeps a symmetric ``[n x n]`` copy for the
    all-pairs gradient kernel."""

    def __init__(self, ys, n_beads):
        super(DistanceErrorModel, self).__init__(ys)
        self.n_beads = int(n_beads)
        self._ymat = {}

    def ymat_device(self, device):
        if device not in self._ymat:
            n = self.n_beads
            y = self.ys.detach().cpu().numpy() if isinstance(self.ys, torch.Tensor) \
                else np.asarray(self.ys, dtype=np.float64)
            m = np.zeros((n, n))
            I, J = np.triu_indices(n, 1)
            m[I, J] = y
            m[J, I] = y
            self._ymat[device] = torch.from_numpy(m).to(device)
        return self._ymat[device]

    def ypacked_device(self, device):
        """The targets in the order the 32..256-bead force kernels hold them
        (binf_pairdist_pack_targets_f64), made once per device; None for other
        bead counts."""
        key = ('packed', device)
        if key not in self._ymat:
            self._ymat[key] = _native.pairdist_pack_targets(self.ymat_device(device))
        return self._ymat[key]

    def clone(self):
        copy = self.__class__(self.ys, self.n_beads)
        copy._dev = self._dev
        copy._ymat = self._ymat
        copy.set_fixed_variables_from_pdf(self)
        return copy

    def native_spec(self):
        from binf_amd.example.likelihood import _unchanged
        if _unchanged(self, GaussianErrorModel, ('_evaluate_log_prob', '_evaluate_gradient')):
            return ('gaussian_pairdist', self)
        return None


def make_distance_likelihood(target_distances, n_beads):
    from binf_amd.pdf.likelihoods import Likelihood
    return Likelihood('restraints', DistanceForwardModel(n_beads),
                      DistanceErrorModel(target_distances, n_beads))


def native_log_prob(likelihood, fwm, em, fwm_vars, em_vars):
    """Fused log-likelihood for the (DistanceForwardModel, DistanceErrorModel)
    pair -- the same bits as forward model + error model, without the
    ``[C x n_pairs]`` distances in HBM; None if the inputs are not device
    tensors."""
    fwm_vars, em_vars = dict(fwm_vars), dict(em_vars)
    fwm._complete_variables(fwm_vars)
    em._complete_variables(em_vars)
    x = fwm_vars.get('coordinates')
    if not (isinstance(x, torch.Tensor) and x.is_cuda) or 'precision' not in em_vars:
        return None
    x2 = x if x.dim() == 2 else x.reshape(1, -1)
    I, J = fwm.pair_index(x.device)
    x2 = x2.contiguous()
    ys = em.ys_device(x.device)
    if USE_CHI2_MEMO and ys.numel() >= 2048 and x2.numel() * 8 <= (1 << 28):
        # HMCSampler.sample() asks for the log-prob of the state it ended the last
        # transition with again as E_before: a per-chain memo of chi^2, checked on the
        # device bit for bit (include/binf_hip.h, binf_pairdist_gauss_logp_memo_f64)
        return _native.pairdist_gauss_logp_memo(x2, I, J, ys, em_vars['precision'],
                                                _chi2_memo(I, ys, x2.shape))
    return _native.pairdist_gauss_logp(x2, I, J, ys, em_vars['precision'])


def make_restraint_gibbs_sampler(posterior, timestep, nsteps, start_state, **hmc_kwargs):
    """Gibbs-within-HMC for the restraint posterior, wired like the example's
    ``make_hmc_sampler`` (reference scheme ``binf/example/samplers.py:94-111``): HMC on the
    coordinates given the precision, the conjugate Gamma draw of the precision given the
    coordinates (one precision per chain).  ``posterior`` holds the restraint likelihood, a
    prior on ``coordinates`` and a :class:`GammaPrior` on ``precision``."""
    from binf_amd.example.samplers import GammaSampler
    from binf_amd.samplers.gibbs import GibbsSampler
    from binf_amd.samplers.hmc import HMCSampler

    class RestraintPrecisionSampler(GammaSampler):
        """The example's GammaSampler with the restraint likelihood's names."""

        def _calculate_shape(self):
            n = len(self.pdf.likelihoods['restraints'].error_model.ys)
            return 0.5 * n + self._get_prior().shape - 1

        def _unit_precision_log_prob(self):
            return self.pdf.likelihoods['restraints'].log_prob(
                coordinates=self.pdf['coordinates'].value, precision=1.0)

    coords = start_state.variables['coordinates']
    precision = start_state.variables['precision']
    rng = hmc_kwargs.get('rng')
    coords_sampler = HMCSampler(posterior.conditional_factory(precision=precision), coords, timestep,
                                nsteps, variable_name='coordinates', **hmc_kwargs)
    precision_sampler = RestraintPrecisionSampler(
        posterior.conditional_factory(coordinates=coords), precision,
        rng=rng if hasattr(rng, 'gamma') else None)
    return GibbsSampler(posterior, start_state, {'coordinates': coords_sampler,
                                                 'precision': precision_sampler})


def native_hmc_energy(likelihood, x2, p2, precision, prior, terms):
    """``0.5 * sum(p**2) - log_prob`` of a posterior made of this likelihood, at most one
    isotropic Gaussian prior ``(k, x0)`` and up to two constants of the move, added in the
    order of ``terms`` (``Posterior.native_energy_spec``), one launch:
    binf_pairdist_hmc_energy_f64.  Shares the chi^2 memo with ``native_log_prob``."""
    fwm, em = likelihood.forward_model, likelihood.error_model
    I, J = fwm.pair_index(x2.device)
    ys = em.ys_device(x2.device)
    memo = None
    if USE_CHI2_MEMO and ys.numel() >= 2048 and x2.numel() * 8 <= (1 << 28):
        memo = _chi2_memo(I, ys, x2.shape)
    return _native.pairdist_hmc_energy(x2, p2, I, J, ys, precision, prior, False, memo, terms=terms)


USE_CHI2_MEMO = True


def _chi2_memo(I, ys, shape):
    from binf_amd import memo
    return memo.chi2_memo(ys, I, shape, ys.device, _native.new_chi2_memo)


def native_gradient(likelihood, fwm, em, fwm_vars, em_vars):
    """Fused all-pairs force for the (DistanceForwardModel, DistanceErrorModel)
    pair; None if the inputs are not device tensors."""
    fwm_vars, em_vars = dict(fwm_vars), dict(em_vars)
    fwm._complete_variables(fwm_vars)
    em._complete_variables(em_vars)
    x = fwm_vars.get('coordinates')
    if not (isinstance(x, torch.Tensor) and x.is_cuda) or 'precision' not in em_vars:
        return None
    x2 = x if x.dim() == 2 else x.reshape(1, -1)
    out = _native.pairdist_gauss_grad(x2.contiguous(), em.ymat_device(x.device),
                                      em_vars['precision'], packed=em.ypacked_device(x.device))
    return out if x.dim() == 2 else out.reshape(-1)


# ---------------------------------------------------------------------------
# recognition of the restraint posterior and the hooks of the kind (binf_amd/native.py)
# ---------------------------------------------------------------------------
def _posterior_spec(posterior, variable_name, strict):
    """Recognised: exactly one restraint likelihood (pair-distance forward model +
    Gaussian error model with the precision fixed) plus at most one isotropic
    Gaussian prior on the same variable; components without a differentiable
    variable do not enter the force anyway (quirk Q4).  The result records the
    order of the two force terms, which is the Posterior's sorted-component-name
    order.

    ``strict`` (the energy, where EVERY component counts: a component without
    differentiable variables drops out of the force, not out of ``log_prob``):
    besides the likelihood and the one prior, up to two components whose variables
    are ALL fixed (constants of the move, e.g. the GammaPrior of the precision
    inside a Gibbs sweep) are recorded in place -- the last entry lists ``'prior'`` /
    ``'lik'`` / such a component in the Posterior's order."""
    from binf_amd.pdf import IsotropicGaussian
    from binf_amd.pdf.likelihoods import Likelihood
    lik = prior = None
    order = []
    terms = []          # strict: every component in the Posterior's order
    for f in posterior._ordered_components():
        if not (len(f.variables) > 0 and len(f.differentiable_variables) > 0):
            if strict:
                # out of the force, not out of log_prob: a component with every variable
                # fixed is a constant of the move the energy kernel can add in its place
                if len(f.variables) > 0 or sum(1 for t in terms if not isinstance(t, str)) == 2:
                    return None
                terms.append(f)
            continue
        if isinstance(f, Likelihood):
            fs = getattr(f.forward_model, 'native_spec', lambda: None)()
            es = getattr(f.error_model, 'native_spec', lambda: None)()
            if lik is not None or fs is None or es is None or \
                    fs[0] != 'pairdist' or es[0] != 'gaussian_pairdist' or \
                    f.variables != {variable_name} or \
                    'precision' not in es[1].parameters:
                return None
            lik = f
            order.append('lik')
            terms.append('lik')
        elif isinstance(f, IsotropicGaussian):
            if prior is not None or f.native_hmc_spec(variable_name) is None:
                return None
            prior = f
            order.append('prior')
            terms.append('prior')
        else:
            return None
    if lik is None:
        return None
    em = lik.error_model
    params = (em, em['precision'].value,
              None if prior is None else (float(prior['k'].value), float(prior['x0'].value)),
              order[0] == 'prior')
    return params + (lik, terms) if strict else params


def match_leapfrog(posterior, variable_name):
    return _posterior_spec(posterior, variable_name, strict=False)


def match_energy(posterior, variable_name):
    return _posterior_spec(posterior, variable_name, strict=True)


def leapfrog(sampler, spec, q2, p2, dt, dtc, nsteps, mode, q_from):
    """The whole integration in one launch (bit-identical to the per-step loop)."""
    _, em, precision, prior, prior_first = spec
    packed = getattr(em, 'ypacked_device', None)
    n = q2.shape[1] // 3
    # beyond 1024 beads only as a wave per tile: packed targets + the library's workspace
    if q2.shape[1] % 3 != 0:
        return False
    if n > 1024:
        pk = packed(q2.device) if packed is not None else None
        if pk is None or _native._pairdist_tiles_workspace(q2.shape[0], n, pk, q2.device)[1] <= 0:
            return False                   # no packed form / no room for the scratch: the per-step tier
    qf = None
    if q_from is not None:
        qf = q_from
        if not (qf.is_contiguous() and qf.shape == q2.shape and qf.dtype == q2.dtype
                and qf.device == q2.device):
            q2.copy_(qf)
            qf = None
    _native.pairdist_leapfrog(q2, p2, em.ymat_device(q2.device), precision,
                              prior, prior_first, dt, dtc, nsteps, mode,
                              packed=packed(q2.device) if packed is not None else None,
                              q_from=qf)
    return True


def energy(sampler, spec, q0):
    """``E(x, momentum)`` with the prior row sum, chi^2 (with its memo), term sum and
    kinetic energy in one launch, or None (a term the kernel cannot take)."""
    if q0.shape[1] % 3 != 0 or q0.shape[1] // 3 > 2048:
        return None
    _, em, precision, prior, prior_first, lik, terms = spec
    # constants of the move (components with every variable fixed): once per sample()
    terms = [t if isinstance(t, str) else t.log_prob() for t in terms]

    def kernel_term(t):
        if isinstance(t, str):
            return True
        if isinstance(t, torch.Tensor):
            return t.dim() == 0 or (t.is_cuda and t.dtype == torch.float64 and
                                    t.is_contiguous() and t.numel() == q0.shape[0])
        try:                                # a Python / numpy scalar
            float(t)
            return True
        except (TypeError, ValueError):
            return False
    if not all(kernel_term(t) for t in terms):
        return None
    return lambda x, mom: native_hmc_energy(lik, x, mom, precision, prior, terms)


native.register(
    KIND, replace=True,
    match_leapfrog=match_leapfrog, match_energy=match_energy,
    leapfrog=leapfrog, energy=energy,
    likelihood={('pairdist', 'gaussian_pairdist'): (native_log_prob, native_gradient)})

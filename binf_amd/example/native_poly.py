"""
Native fast paths of the polynomial + Gaussian likelihood: when a
``Likelihood`` is built from :class:`ForwardModel` (with numpy's ``polyval``)
and :class:`GaussianErrorModel`, its log-prob and gradient run in fused HIP
kernels and the ``[C x n_data]`` mock data is never written to HBM
(reference path: ``binf/pdf/likelihoods.py:141-155`` calling
``binf/example/likelihood.py:24-30,54-61``).
"""
import torch

from binf_amd import _native


def _usable(coeffs):
    return isinstance(coeffs, torch.Tensor) and coeffs.is_cuda and \
        coeffs.dtype == torch.float64 and coeffs.shape[-1] <= 64


def _as2d(x):
    return x if x.dim() == 2 else x.reshape(1, -1)


def log_prob(likelihood, pair, fwm_vars, em_vars):
    (_, fwm), (_, em) = pair
    fwm_vars = dict(fwm_vars)
    em_vars = dict(em_vars)
    fwm._complete_variables(fwm_vars)
    em._complete_variables(em_vars)
    coeffs = fwm_vars.get('coefficients')
    if not _usable(coeffs) or 'precision' not in em_vars:
        return None
    dev = coeffs.device
    return _native.poly_gauss_logp(_as2d(coeffs).contiguous(), fwm.xs_device(dev),
                                   em.ys_device(dev), em_vars['precision'])


def gradient(likelihood, pair, fwm_vars, em_vars):
    (_, fwm), (_, em) = pair
    fwm_vars = dict(fwm_vars)
    em_vars = dict(em_vars)
    fwm._complete_variables(fwm_vars)
    em._complete_variables(em_vars)
    coeffs = fwm_vars.get('coefficients')
    if not _usable(coeffs) or 'precision' not in em_vars:
        return None
    dev = coeffs.device
    c2 = _as2d(coeffs).contiguous()
    out = _native.poly_gauss_grad(c2, fwm.design_matrix(c2.shape[1], dev),
                                  em.ys_device(dev), em_vars['precision'])
    return out if coeffs.dim() == 2 else out.reshape(-1)


FUSED_MAX_COEFFS = 16       # binf_hmc_sample_poly_f64 limits (include/binf_hip.h)
FUSED_MAX_DATA = 1024        # > 128: one wave per chain (csrc/hmc_poly_wave.hip)


def posterior_hmc_spec(posterior, variable_name):
    """Descriptor of the fused small-data trajectory kernel
    (``binf_hmc_sample_poly_f64``) if ``posterior`` is one it integrates:
    a conditional posterior whose only free variable is ``coefficients``, made
    of ONE polynomial + Gaussian-error likelihood with the precision fixed, at
    most one :class:`GaussianPrior` on the coefficients, and components without
    free variables (constants of the energy) before and / or one after them in
    the posterior's summation order.  Otherwise None (per-step tier).

    ``('poly', forward_model, error_model, precision, prior or None,
    prior_first, [constants before], constant after or None)``"""
    from binf_amd.example.priors import GaussianPrior
    from binf_amd.pdf.likelihoods import Likelihood
    if variable_name != 'coefficients':
        return None
    lik = prior = None
    kinds, consts = [], []
    for f in posterior._ordered_components():
        if len(f.variables) == 0:
            kinds.append('c')
            consts.append(f)
        elif set(f.variables) == {variable_name} and isinstance(f, Likelihood) \
                and lik is None and f._native_pair() is not None \
                and 'precision' in f.error_model.parameters:
            lik = f
            kinds.append('lik')
        elif set(f.variables) == {variable_name} and isinstance(f, GaussianPrior) \
                and prior is None:
            prior = f
            kinds.append('prior')
        else:
            return None
    n_data = len(lik.error_model.ys) if lik is not None else 0
    if lik is None or n_data > FUSED_MAX_DATA or \
            (n_data > 128 and _native.pairwise_tree_height(n_data) > 3):
        return None
    theta = [i for i, k in enumerate(kinds) if k != 'c']
    n_pre, n_post = theta[0], len(kinds) - 1 - theta[-1]
    if theta[-1] - theta[0] + 1 != len(theta) or n_post > 1:
        return None                  # a constant between the two, or two after
    return ('poly', lik.forward_model, lik.error_model,
            lik.error_model['precision'].value, prior,
            prior is not None and kinds.index('prior') < kinds.index('lik'),
            consts[:n_pre], consts[n_pre] if n_post else None)

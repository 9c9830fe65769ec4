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


def posterior_hmc_spec(posterior, variable_name):
    """No fused trajectory kernel for the polynomial posterior yet: HMC runs
    on the generic tier around the fused log-prob / gradient kernels."""
    return None

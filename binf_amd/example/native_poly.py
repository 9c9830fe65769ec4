"""Placeholder until the polynomial kernels land (next commit)."""


def log_prob(likelihood, pair, fwm_vars, em_vars):
    return None


def gradient(likelihood, pair, fwm_vars, em_vars):
    return None


def posterior_hmc_spec(posterior, variable_name):
    return None

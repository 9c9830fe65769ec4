"""Factories of the example application (mirror of reference
``binf/example/misc.py:18-33``; the plotting / prediction helpers are out of
scope)."""
import numpy as np


def get_MAP(samples, log_probs):
    best = samples[int(np.argmax(log_probs))]
    return best.variables['coefficients'], best.variables['precision']


def make_posterior(xses, ys, polynomial):
    from binf_amd.example.likelihood import make_likelihood
    from binf_amd.example.priors import make_priors
    from binf_amd.pdf.posteriors import Posterior
    L = make_likelihood(xses, ys, polynomial)
    return Posterior({L.name: L}, make_priors())

"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  numpy formulation of the
BUILD-DEFINED pairwise-distance-restraint posterior (BASELINE config C5).

The reference holds no code for this model (README.rst:9 only names the
application), so there is nothing to pin it against: **parity unpinned by the
reference**; this file is the definition the HIP kernels are held to.  It is
written in the shape of the reference's example likelihood
(binf/example/likelihood.py:40-68: Gaussian error model; binf/pdf/likelihoods.py:
141-155: log-prob and chain-rule gradient).
"""
import numpy as np


def pairs(n_beads):
    return np.triu_indices(n_beads, 1)


def forward(coords, n_beads):
    """mock data: distances of all bead pairs, np.triu order."""
    x = np.asarray(coords, dtype=np.float64).reshape(n_beads, 3)
    I, J = pairs(n_beads)
    return np.sqrt(np.sum((x[I] - x[J]) ** 2, axis=1))


def log_prob(coords, ys, precision, n_beads):
    mock = forward(coords, n_beads)
    logZ = len(ys) * 0.5 * np.log(precision)
    return -0.5 * np.sum((mock - ys) ** 2) * precision + logZ


def gradient(coords, ys, precision, n_beads):
    """J . ((mock - ys) * precision), J = d mock / d coords, accumulated pair
    by pair (the Jacobian itself is never built)."""
    x = np.asarray(coords, dtype=np.float64).reshape(n_beads, 3)
    I, J = pairs(n_beads)
    diff = x[I] - x[J]
    d = np.sqrt(np.sum(diff ** 2, axis=1))
    w = ((d - ys) * precision / d)[:, None] * diff
    g = np.zeros_like(x)
    np.add.at(g, I, w)
    np.add.at(g, J, -w)
    return g.reshape(-1)


class DistancePosterior(object):
    """Duck-typed pdf for oracle.ref_numpy.RefHMCSampler: restraint likelihood
    (+ optional isotropic Gaussian prior k/2 |x|^2 on the coordinates)."""

    def __init__(self, ys, precision, n_beads, prior_k=0.0, variable_name='coordinates'):
        self.ys, self.precision, self.n_beads = ys, precision, n_beads
        self.prior_k = prior_k
        self.variable_name = variable_name

    def log_prob(self, **v):
        x = v[self.variable_name]
        lp = log_prob(x, self.ys, self.precision, self.n_beads)
        if self.prior_k:
            lp = (-0.5 * self.prior_k * np.sum((x - 0.0) ** 2)) + lp
        return lp

    def gradient(self, **v):
        x = v[self.variable_name]
        g = gradient(x, self.ys, self.precision, self.n_beads)
        if self.prior_k:
            g = self.prior_k * (x - 0.0) + g
        return g

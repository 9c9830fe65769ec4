"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  numpy restatement of the
binf hot path, one chain per sampler object exactly as the reference runs it.

Parity status (see oracle/__init__.py): pinned by the reference's own test
known-answers for the PDF / Likelihood / Gibbs plumbing; ``_leapfrog``
(``binf/samplers/hmc.py:92-125``) pinned bit for bit by outputs of the reference's
own code (``tests/golden/ref_leapfrog_*.npz``, ``oracle/gen_ref_leapfrog.py``);
``sample()`` (``hmc.py:136-164``) pinned statement by statement by
``tests/golden/ref_sample_*.npz`` except the one csb line (``:151``): **parity
unpinned** is only the definition of ``csb.numeric.exp`` (clip bounds below).

Every function cites the reference lines (relative to the reference root) it
restates.  Arithmetic is kept in the reference's operation order so that the
C restatement (oracle/oracle_c.c) and the HIP kernels can be held to it
bit-for-bit.
"""
from collections import namedtuple
from copy import deepcopy

import numpy as np

# --------------------------------------------------------------------------
# third-party boundary: csb.numeric.exp  (used at binf/samplers/hmc.py:10,151)
# --------------------------------------------------------------------------
# CSB toolbox, version unpinned by the reference (setup.py:25).  Restated from
# its published source csb/numeric/__init__.py: exp(x) = numpy.exp(clip(x,
# EXP_MIN, EXP_MAX)) with EXP_MIN = -308, EXP_MAX = +709.  Not verifiable in
# this container -> the bounds are "parity unpinned"; they only matter when
# |dE| exceeds them.
EXP_MIN = -308.0
EXP_MAX = 709.0


def exp(x, x_min=EXP_MIN, x_max=EXP_MAX):
    x_min = max(x_min, EXP_MIN)
    x_max = min(x_max, EXP_MAX)
    return np.exp(np.clip(x, x_min, x_max))


# --------------------------------------------------------------------------
# numpy's pairwise summation (what np.sum does on a contiguous f64 vector);
# the energy reductions at hmc.py:148,150 and pdf/__init__.py:185 go through it
# --------------------------------------------------------------------------
PW_BLOCKSIZE = 128


def pairwise_sum_py(a):
    """Pure-Python restatement of numpy's DOUBLE_pairwise_sum (the loop behind
    np.add.reduce).  Bit-identical to ``np.sum`` up to the outer ``0.0 +``
    (see :func:`np_sum_py`).  Small inputs only -- it is a Python loop."""
    n = len(a)
    if n < 8:
        res = -0.0
        for i in range(n):
            res += a[i]
        return res
    if n <= PW_BLOCKSIZE:
        r = [a[j] for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] += a[i + j]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res += a[i]
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return pairwise_sum_py(a[:n2]) + pairwise_sum_py(a[n2:])


NPY_BUFSIZE = 8192   # numpy's default ufunc buffer, in elements


def np_sum_py(a):
    """np.sum(a) for a 1-D contiguous f64 array: the reduction starts from the
    additive identity +0.0 and adds, one after the other, the pairwise sums of
    consecutive 8192-element chunks (numpy's buffered reduction loop; a vector
    of <= 8192 elements is one chunk)."""
    a = [float(x) for x in a]
    res = 0.0
    if not a:
        return res + pairwise_sum_py(a)
    for i in range(0, len(a), NPY_BUFSIZE):
        res = res + pairwise_sum_py(a[i:i + NPY_BUFSIZE])
    return res


def pairwise_leaves(n):
    """(offset, length) of every leaf block of the pairwise recursion for a
    length-n vector, in order; plus the nesting as a tree of leaf indices.
    Used by tests to check the device kernels' tree walk."""
    leaves = []

    def rec(off, m):
        if m <= PW_BLOCKSIZE:
            leaves.append((off, m))
            return len(leaves) - 1
        n2 = m // 2
        n2 -= n2 % 8
        left = rec(off, n2)
        right = rec(off + n2, m - n2)
        return (left, right)

    tree = rec(0, n)
    return leaves, tree


# --------------------------------------------------------------------------
# HMC sampler: binf/samplers/hmc.py
# --------------------------------------------------------------------------
HMCSampleStats = namedtuple('HMCSampleStats', 'accepted stepsize')  # hmc.py:12


class RefHMCSampler(object):
    """Restatement of ``HMCSampler`` (binf/samplers/hmc.py:15-191).

    ``normal`` / ``uniform`` default to the global legacy numpy stream the
    reference consumes (hmc.py:146,151); tests may inject recorded draws.
    """

    def __init__(self, pdf, state, timestep, nsteps, timestep_adaption_limit=0,
                 adaption_uprate=1.05, adaption_downrate=0.95,
                 variable_name=None, normal=None, uniform=None):
        # hmc.py:51-62
        self.pdf = pdf
        self.state = state
        self.timestep = timestep
        self.nsteps = nsteps
        self.timestep_adaption_limit = timestep_adaption_limit
        self.adaption_uprate = adaption_uprate
        self.adaption_downrate = adaption_downrate
        self._variable_name = variable_name
        self._last_move_accepted = 0
        self.n_accepted = 0
        self.counter = 0
        self._normal = normal if normal is not None else \
            (lambda size: np.random.normal(size=size))
        self._uniform = uniform if uniform is not None else \
            (lambda: np.random.uniform())
        # extras recorded for the golden vectors (not in the reference)
        self.last_E_before = None
        self.last_E_after = None

    @property
    def acceptance_rate(self):  # hmc.py:64-69
        if self.counter > 0:
            return self.n_accepted / float(self.counter)
        return 0.0

    @property
    def variable_name(self):  # hmc.py:71-80
        return 'HMC' if self._variable_name is None else self._variable_name

    @property
    def last_move_accepted(self):  # hmc.py:82-90
        return self._last_move_accepted

    def _leapfrog(self, q, p, timestep, nsteps):  # hmc.py:92-125
        gradient = lambda x: self.pdf.gradient(**{self._variable_name: x})
        p -= 0.5 * timestep * gradient(q)            # :116
        for i in range(nsteps - 1):                  # :118
            q += p * timestep                        # :119
            p -= timestep * gradient(q)              # :120
        q += p * timestep                            # :122
        p -= 0.5 * timestep * gradient(q)            # :123
        return q, p

    def sample(self):  # hmc.py:136-164
        V = lambda x: -self.pdf.log_prob(**{self._variable_name: x})   # :143
        q = deepcopy(self.state)                                       # :145
        p = self._normal(q.shape)                                      # :146
        E_before = V(q) + 0.5 * np.sum(p ** 2)                         # :148
        q, p = self._leapfrog(q, p, self.timestep, self.nsteps)        # :149
        E_after = V(q) + 0.5 * np.sum(p ** 2)                          # :150
        acc = self._uniform() < exp(-(E_after - E_before))             # :151
        self.last_E_before, self.last_E_after = E_before, E_after
        self._last_move_accepted = acc                                 # :153
        self.counter += 1                                              # :154
        if self.counter < self.timestep_adaption_limit:                # :156
            self._adapt_timestep()
        if acc:                                                        # :159
            self.state = q
            self.n_accepted += 1
            return deepcopy(q)
        return deepcopy(self.state)                                    # :164

    @property
    def last_draw_stats(self):  # hmc.py:166-181
        return {self.variable_name: HMCSampleStats(self.last_move_accepted,
                                                   self.timestep)}

    def _adapt_timestep(self):  # hmc.py:183-191 (uprate on ACCEPT: quirk Q3)
        if self.last_move_accepted:
            self.timestep *= self.adaption_uprate
        else:
            self.timestep *= self.adaption_downrate


# --------------------------------------------------------------------------
# PDFs, duck-typed to the sampler's contract (log_prob(**kw), gradient(**kw))
# --------------------------------------------------------------------------
class GaussianPDF(object):
    """The reference's own isotropic Gaussian, ``TestHO``
    (binf/pdf/__init__.py:163-191): log p = -0.5*k*sum((x-x0)**2),
    gradient = k*(x-x0)  [gradient of the ENERGY, -log p]."""

    def __init__(self, k=1.0, x0=0.0, variable_name='x'):
        self.k = k
        self.x0 = x0
        self.variable_name = variable_name

    def log_prob(self, **variables):   # pdf/__init__.py:181-185
        x = variables[self.variable_name]
        return -0.5 * self.k * np.sum((x - self.x0) ** 2)

    def gradient(self, **variables):   # pdf/__init__.py:187-191
        x = variables[self.variable_name]
        return self.k * (x - self.x0)


def polyval(x, c):
    """numpy.polynomial.polynomial.polyval restated for a 1-D coefficient
    vector (the ``polynomial`` callable of example_script.py:21): Horner from
    the highest coefficient, ``c0 = c[-i] + c0*x``."""
    c = np.asarray(c, dtype=np.float64)
    c0 = c[-1] + x * 0
    for i in range(2, len(c) + 1):
        c0 = c[-i] + c0 * x
    return c0


class PolyCoefficientsConditional(object):
    """What HMC sees when it samples ``coefficients`` of the example posterior
    with ``precision`` fixed (the conditional Posterior GibbsSampler installs,
    binf/samplers/gibbs.py:50-52).

    log_prob: sum over ALL components, in sorted-component-name order (the
      build's documented resolution of quirk Q5; binf/pdf/posteriors.py:147-151):
      'coefficients_prior' -> GaussianPrior   -0.5*sum((c-means)**2/variances)
                                               (binf/example/priors.py:49-54)
      'points'             -> Likelihood      -0.5*sum((mock-ys)**2)*tau
                                               + len(ys)*0.5*log(tau)
                                               (binf/example/likelihood.py:54-57)
      'precision_prior'    -> GammaPrior      (shape-1)*log(tau) - tau*rate
                                               (binf/example/priors.py:23-25)
    gradient: ONLY the likelihood term J . (mock-ys)*tau -- the priors have no
      differentiable variable and are skipped (quirk Q4,
      binf/pdf/posteriors.py:183; binf/pdf/likelihoods.py:148-155;
      binf/example/likelihood.py:28-30,59-61).
    """

    def __init__(self, xses, ys, precision, prior_means, prior_variances,
                 gamma_shape, gamma_rate, variable_name='coefficients'):
        self.xses = np.asarray(xses, dtype=np.float64)
        self.ys = np.asarray(ys, dtype=np.float64)
        self.precision = precision
        self.means = np.asarray(prior_means, dtype=np.float64)
        self.variances = np.asarray(prior_variances, dtype=np.float64)
        self.gamma_shape = gamma_shape
        self.gamma_rate = gamma_rate
        self.variable_name = variable_name

    def component_log_probs(self, c):
        tau = self.precision
        mock = polyval(self.xses, c)
        logZ = len(self.ys) * 0.5 * np.log(tau)
        lik = -0.5 * np.sum((mock - self.ys) ** 2) * tau + logZ
        cprior = -0.5 * np.sum((c - self.means) ** 2 / self.variances)
        pprior = (self.gamma_shape - 1.0) * np.log(tau) - tau * self.gamma_rate
        return {'coefficients_prior': cprior, 'points': lik,
                'precision_prior': pprior}

    def log_prob(self, **variables):
        comps = self.component_log_probs(variables[self.variable_name])
        # numpy.sum over a short Python list: sequential, sorted-name order
        return np.sum([comps[k] for k in sorted(comps)])

    def jacobi_matrix(self, c):   # binf/example/likelihood.py:28-30
        return np.vstack([self.xses ** i for i in range(len(c))])

    def gradient(self, **variables):
        c = variables[self.variable_name]
        mock = polyval(self.xses, c)
        dfm = self.jacobi_matrix(c)
        emgrad = (mock - self.ys) * self.precision
        return dfm.dot(emgrad)    # binf/pdf/likelihoods.py:155


# --------------------------------------------------------------------------
# conjugate precision update: binf/example/samplers.py:7-51
# --------------------------------------------------------------------------
def gamma_shape(n_data, prior_shape):
    """GammaSampler._calculate_shape (binf/example/samplers.py:27-32):
    0.5*n + prior.shape - 1  (as written; one less than the textbook value)."""
    return 0.5 * n_data + prior_shape - 1


def gamma_rate(xses, ys, coefficients, prior_rate):
    """GammaSampler._calculate_rate (binf/example/samplers.py:34-41):
    -L.log_prob(coefficients, precision=1.0) + prior.rate, with
    L.log_prob(.., precision=1) = -0.5*sum((mock-ys)**2)*1.0 + n*0.5*log(1.0)
    (binf/example/likelihood.py:54-57)."""
    mock = polyval(xses, coefficients)
    logZ = len(ys) * 0.5 * np.log(1.0)
    r1 = -(-0.5 * np.sum((mock - ys) ** 2) * 1.0 + logZ)
    return r1 + prior_rate


def gamma_draw(g, rate):
    """GammaSampler.sample (binf/example/samplers.py:43-51):
    np.random.gamma(shape) / rate, with the variate g supplied."""
    return g / rate


# --------------------------------------------------------------------------
# batched driver used by tests / bench cpu_baseline: C independent reference
# samplers, one chain each, draws injected
# --------------------------------------------------------------------------
def hmc_sample_batch(make_pdf, q0, p0, u, timestep, nsteps, variable_name='x',
                     adapt=False, uprate=1.05, downrate=0.95):
    """Run one ``sample()`` per chain with injected momentum ``p0[c]`` and
    uniform ``u[c]``.  ``timestep`` is a scalar or a length-C vector.
    Returns dict(q_out, accepted, e_before, e_after, timestep_out)."""
    q0 = np.asarray(q0, dtype=np.float64)
    C, D = q0.shape
    dt = np.broadcast_to(np.asarray(timestep, dtype=np.float64), (C,)).copy()
    q_out = np.empty_like(q0)
    acc = np.zeros(C, dtype=np.uint8)
    eb = np.empty(C)
    ea = np.empty(C)
    for c in range(C):
        s = RefHMCSampler(make_pdf(c), q0[c].copy(), float(dt[c]), nsteps,
                          timestep_adaption_limit=2 if adapt else 0,
                          adaption_uprate=uprate, adaption_downrate=downrate,
                          variable_name=variable_name,
                          normal=lambda size, c=c: p0[c].copy(),
                          uniform=lambda c=c: u[c])
        q_out[c] = s.sample()
        acc[c] = 1 if s.last_move_accepted else 0
        eb[c] = s.last_E_before
        ea[c] = s.last_E_after
        dt[c] = s.timestep
    return dict(q_out=q_out, accepted=acc, e_before=eb, e_after=ea,
                timestep_out=dt)

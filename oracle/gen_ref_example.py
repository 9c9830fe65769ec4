"""
TEST INFRASTRUCTURE ONLY.  Writes tests/golden/ref_example_*.npz: outputs of the REFERENCE'S OWN
example-model and subsampler code (``binf/example/likelihood.py``, ``priors.py``, ``samplers.py``),
run in the build container.

PROVENANCE -- read this.  The reference's example classes cannot be instantiated: their base classes
come from the absent third-party ``csb`` toolbox (nothing is substituted for it), and
``binf/example/samplers.py`` is Python-2 source (a ``print`` statement, ``filter(...)[0]``).  What
CAN run, unchanged, is the arithmetic these classes are made of:

* **method bodies** -- the function definitions ``ForwardModel._evaluate`` /
  ``_evaluate_jacobi_matrix`` (``likelihood.py:24-30``), ``GaussianErrorModel._evaluate_log_prob`` /
  ``_evaluate_gradient`` (``:54-61``), ``GammaPrior._evaluate_log_prob`` (``priors.py:23-25``) and
  ``GaussianPrior._evaluate_log_prob`` (``:49-54``) are taken out of the reference's syntax tree,
  compiled as they stand and called with a DATA-ONLY ``self`` (an object holding ``xses``,
  ``polynomial``, ``ys``, ``shape``, ``rate``, ``{'means': ..., 'variances': ...}`` -- the attributes
  those bodies read; no behaviour);
* **``RWMCSampler``** (``samplers.py:54-92``): a plain ``object`` class without any csb import -- its
  source lines are executed as they are and ``sample()`` is called on a real instance;
* **``GammaSampler._calculate_shape`` / ``_calculate_rate`` / ``sample``** (``samplers.py:27-51``),
  executed as they are; the method they call first, ``_get_prior`` (``:14-25``), is the Python-2-only
  part (``print prior``, ``filter(...)[0]``) and is the ONE method replaced here: by a subclass method
  that returns the precision prior directly -- which is what the reference's method does after
  printing it.

The glue is the reference's own wherever it is free of csb and of Python-2-only calls: the method
bodies of ``Likelihood._split_variables / _evaluate_log_prob / _evaluate_gradient``
(``binf/pdf/likelihoods.py:122-155``) and of the Posterior's log-prob
(``_get_component_variables_list / _evaluate_components / _evaluate_log_prob``,
``posteriors.py:117-151``) are compiled unchanged as well and run over duck-typed models /
components.  What this script supplies itself: the ORDER of the Posterior's components (sorted
name -- quirk Q5: the reference's is Python-2 hash order), the completion of fixed variables
(``binf/pdf/__init__.py:153-160``), and the Gibbs sweep (``gibbs.py:136-151``: alphabetical,
conditionals refreshed from the state; quirk Q6: the conditional copy of the GammaPrior carries
rate = shape) -- ``GibbsSampler`` itself derives from a csb class.  Those stay pinned by the
reference's unit-test known answers.

Pinned by these fixtures (see tests/test_ref_example.py): Horner via the callable of
``example_script.py:21``, the design matrix, the Gaussian error model's log-prob and gradient in the
reference's operation order, both priors, the conjugate shape / rate / draw (incl. the ``- 1`` of
``samplers.py:32``), the random-walk Metropolis move with numpy's ``exp`` and its stream order.

Run (build container only):   python -m oracle.gen_ref_example
"""
import ast
import os
import re
import textwrap

import numpy as np

REF = '/root/reference/binf/example'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')

PROVENANCE = ('outputs of the REFERENCE\'s example code executed from its own source: method bodies '
              'of binf/example/likelihood.py:24-30,54-61 and priors.py:23-25,49-54 (function '
              'definitions compiled unchanged, called with a data-only self), class RWMCSampler '
              '(samplers.py:54-92, unchanged) and GammaSampler._calculate_shape/_calculate_rate/'
              'sample (samplers.py:27-51, unchanged; the Python-2-only _get_prior replaced by a '
              'method returning the prior), glued by the method bodies of Likelihood '
              '(binf/pdf/likelihoods.py:122-155) and of the Posterior log-prob (posteriors.py:117-151), '
              'unchanged, over duck-typed components; csb absent, nothing substituted for it; the '
              'component order (sorted name, Q5), the completion of fixed variables and the Gibbs '
              'sweep supplied by oracle/gen_ref_example.py; numpy %s' % np.__version__)


def method(path, cls, name):
    """The function ``cls.name`` of the reference file ``path``, compiled from its syntax tree
    alone (the class statement itself is never executed: its bases need csb)."""
    tree = ast.parse(open(path).read(), path)
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name == name:
                    mod = ast.Module(body=[item], type_ignores=[])
                    ns = {'np': np, 'numpy': np}
                    exec(compile(mod, '%s:%s.%s' % (path, cls, name), 'exec'), ns)
                    return ns[name]
    raise LookupError('%s.%s not found in %s' % (cls, name, path))


def class_source(path, cls, drop_methods=()):
    """Source text of the top-level class ``cls`` (the file as a whole is Python-2 source and does
    not parse), optionally without some of its methods."""
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'class %s\b' % cls, l))
    end = next((i for i in range(start + 1, len(lines)) if re.match(r'(class|def)\s', lines[i])), len(lines))
    body = lines[start:end]
    for m in drop_methods:
        s = next(i for i, l in enumerate(body) if re.match(r'\s+def %s\(' % m, l))
        e = next((i for i in range(s + 1, len(body)) if re.match(r'\s+(def|@)', body[i]) and
                  len(body[i]) - len(body[i].lstrip()) == len(body[s]) - len(body[s].lstrip())), len(body))
        body = body[:s] + body[e:]
    return textwrap.dedent('\n'.join(body))


class Data(dict):
    """A data-only ``self``: attributes and ``self['name'].value`` items, nothing else."""

    def __init__(self, items=None, **attrs):
        super(Data, self).__init__(items or {})
        self.__dict__.update(attrs)


class Value(object):
    def __init__(self, value):
        self.value = value


def main():
    lik_py, pri_py, smp_py = (os.path.join(REF, f) for f in ('likelihood.py', 'priors.py', 'samplers.py'))
    fwm_eval = method(lik_py, 'ForwardModel', '_evaluate')
    fwm_jac = method(lik_py, 'ForwardModel', '_evaluate_jacobi_matrix')
    em_logp = method(lik_py, 'GaussianErrorModel', '_evaluate_log_prob')
    em_grad = method(lik_py, 'GaussianErrorModel', '_evaluate_gradient')
    gamma_logp = method(pri_py, 'GammaPrior', '_evaluate_log_prob')
    gauss_logp = method(pri_py, 'GaussianPrior', '_evaluate_log_prob')
    polynomial = np.polynomial.polynomial.polyval            # example_script.py:21
    # ... and the glue they sit in, as far as it is free of csb and of Python-2-only calls: the
    # method bodies of Likelihood (binf/pdf/likelihoods.py:122-155) and of the Posterior's log-prob
    # (binf/pdf/posteriors.py:117-151), given duck-typed components
    pdf_lik_py, post_py = '/root/reference/binf/pdf/likelihoods.py', '/root/reference/binf/pdf/posteriors.py'

    class FwmDuck(object):
        variables = {'coefficients'}

        def __init__(self, data):
            self.data = data

        def __call__(self, coefficients):
            return fwm_eval(self.data, coefficients)

        def jacobi_matrix(self, coefficients):
            return fwm_jac(self.data, coefficients)

    class EmDuck(object):
        variables = {'mock_data', 'precision'}

        def __init__(self, data):
            self.data = data
            self.ys = data.ys

        def log_prob(self, mock_data, precision):
            return em_logp(self.data, mock_data, precision)

        def gradient(self, mock_data, precision):
            return em_grad(self.data, mock_data, precision)

    class RefLikelihood(object):
        """The reference's Likelihood arithmetic: its own _split_variables / _evaluate_log_prob /
        _evaluate_gradient bodies around duck-typed models."""
        _split_variables = method(pdf_lik_py, 'Likelihood', '_split_variables')
        _evaluate_log_prob = method(pdf_lik_py, 'Likelihood', '_evaluate_log_prob')
        _evaluate_gradient = method(pdf_lik_py, 'Likelihood', '_evaluate_gradient')
        variables = {'coefficients', 'precision'}

        def __init__(self, fwm, em):
            self.forward_model, self.error_model = fwm, em

        def log_prob(self, **variables):                 # AbstractBinfPDF.log_prob minus csb's bookkeeping
            return self._evaluate_log_prob(**variables)

        def gradient(self, **variables):
            return self._evaluate_gradient(**variables)

    class RefPosteriorSum(object):
        """The reference's Posterior log-prob arithmetic: its own _get_component_variables_list /
        _evaluate_components / _evaluate_log_prob bodies over duck-typed components held in
        sorted-name order (Q5: the reference's own order is Python-2 hash order)."""
        _get_component_variables_list = method(post_py, 'Posterior', '_get_component_variables_list')
        _evaluate_components = method(post_py, 'Posterior', '_evaluate_components')
        _evaluate_log_prob = method(post_py, 'Posterior', '_evaluate_log_prob')

        def __init__(self, components):
            from collections import OrderedDict
            self._components = OrderedDict((k, components[k]) for k in sorted(components))

    os.makedirs(OUT, exist_ok=True)
    written = []

    # ---- (a) the model arithmetic ------------------------------------------------------------
    for tag, K, N, xlim, C, seed in [('k4_n20', 4, 20, 2.0, 6, 1), ('k7_n37', 7, 37, 1.5, 5, 2),
                                     ('k33_n1000', 33, 1000, 1.0, 4, 3), ('k33_n16384', 33, 16384, 1.0, 2, 4),
                                     ('k5_n8200', 5, 8200, 1.0, 3, 5)]:
        rs = np.random.RandomState(seed)
        xs = np.linspace(-xlim, xlim, N)
        c_true = rs.standard_normal(K)
        ys = polynomial(xs, c_true) + rs.standard_normal(N) / np.sqrt(2.5)
        theta = c_true + 0.3 * rs.standard_normal((C, K))
        taus = rs.uniform(0.5, 4.0, size=C)
        fwm = Data(xses=xs, polynomial=polynomial)
        em = Data(ys=ys)
        means, variances = rs.standard_normal(K), rs.uniform(0.5, 5.0, size=K)
        gp = Data({'means': Value(means), 'variances': Value(variances)})
        gam = Data(shape=1.0, rate=0.2)
        mock = np.stack([fwm_eval(fwm, theta[c]) for c in range(C)])
        jac = fwm_jac(fwm, theta[0])
        rlik = RefLikelihood(FwmDuck(fwm), EmDuck(em))
        lp = np.array([rlik.log_prob(coefficients=theta[c], precision=taus[c]) for c in range(C)])
        lp1 = np.array([rlik.log_prob(coefficients=theta[c], precision=1.0) for c in range(C)])
        eg = np.stack([em_grad(em, mock[c], taus[c]) for c in range(C)])
        grad = np.stack([rlik.gradient(coefficients=theta[c], precision=taus[c]) for c in range(C)])  # likelihoods.py:148-155
        path = os.path.join(OUT, 'ref_example_models_%s.npz' % tag)
        np.savez_compressed(
            path, provenance=np.array(PROVENANCE), xs=xs, ys=ys, theta=theta, precision=taus,
            mock=mock if N <= 1000 else mock[:, ::97], mock_stride=np.int64(1 if N <= 1000 else 97),
            jacobi=jac if K * N <= 40000 else jac[:, ::97], error_logp=lp, error_logp_unit_precision=lp1,
            error_grad=eg if N <= 1000 else eg[:, ::97], likelihood_grad=grad,
            prior_means=means, prior_variances=variances,
            gaussian_prior_logp=np.array([gauss_logp(gp, theta[c]) for c in range(C)]),
            gamma_prior_logp=np.array([gamma_logp(gam, taus[c]) for c in range(C)]),
            gamma_prior_shape=np.float64(1.0), gamma_prior_rate=np.float64(0.2))
        written.append(os.path.basename(path))

    # ---- (b) the example's two subsamplers inside its Gibbs sweep -------------------------------
    ns = {'np': np}
    exec('from collections import namedtuple\n'
         "RWMCSampleStats = namedtuple('RWMCSampleStats', 'acceptance_rate')\n", ns)       # samplers.py:2-4
    exec(compile(class_source(smp_py, 'RWMCSampler'), smp_py + ':RWMCSampler', 'exec'), ns)
    exec(compile(class_source(smp_py, 'GammaSampler', drop_methods=('_get_prior',)),
                 smp_py + ':GammaSampler', 'exec'), ns)
    RWMCSampler, GammaSamplerBase = ns['RWMCSampler'], ns['GammaSampler']

    class GammaSampler(GammaSamplerBase):
        def _get_prior(self):                   # stands for samplers.py:14-25 (Python 2 only)
            return self.pdf.priors['precision_prior']

    class Component(object):
        """A component of a conditional posterior as the Posterior's loop sees it: its free
        variables and log_prob(**free); fixed variables are completed from the conditional's
        current values (AbstractBinfPDF._complete_variables, binf/pdf/__init__.py:153-160)."""

        def __init__(self, variables, fn):
            self.variables, self._fn = variables, fn

        def log_prob(self, **free):
            return self._fn(**free)

    class ConditionalPosterior(dict):
        """The conditional posteriors GibbsSampler installs (gibbs.py:40-52), as far as the two
        subsamplers look at them: log_prob(coefficients=...) through the reference's Posterior
        arithmetic, .likelihoods, .priors, ['coefficients'].value."""

        def __init__(self, lik, cprior, pprior):
            super(ConditionalPosterior, self).__init__()
            self.likelihoods = {'points': lik}
            self.priors = {'coefficients_prior': cprior, 'precision_prior': pprior}
            self.precision = None
            self._sum = RefPosteriorSum({
                'coefficients_prior': Component({'coefficients'}, lambda coefficients: gauss_logp(cprior, coefficients)),
                'points': Component({'coefficients'}, lambda coefficients: lik.log_prob(
                    coefficients=coefficients, precision=self.precision)),
                'precision_prior': Component(set(), lambda: gamma_logp(pprior, self.precision))})

        def log_prob(self, coefficients):
            return self._sum._evaluate_log_prob(coefficients=coefficients)

    for tag, seed, sweeps, stepsize, N in [('seed0', 0, 300, 0.1, 20), ('seed7_n50', 7, 120, 0.05, 50)]:
        np.random.seed(seed)
        real = np.array([2.0, -4.0, 1.0, 1.5])
        xs = np.linspace(-2, 2, N)
        ys = np.random.normal(loc=polynomial(xs, real), scale=1.0 / np.sqrt(2.5))      # example_script.py:22-23
        lik = RefLikelihood(FwmDuck(Data(xses=xs, polynomial=polynomial)), EmDuck(Data(ys=ys)))
        cprior = Data({'means': Value(np.zeros(4)), 'variances': Value(np.ones(4) * 5)})  # priors.py:70
        # Q6: the conditional copies are made by GammaPrior.clone -> (shape, shape)
        pprior = Data(shape=1.0, rate=1.0)
        cond_c = ConditionalPosterior(lik, cprior, pprior)
        cond_p = ConditionalPosterior(lik, cprior, pprior)
        state = {'coefficients': np.ones(4), 'precision': 1.0}                         # example_script.py:25
        rw = RWMCSampler(cond_c, state['coefficients'], stepsize)
        gs = GammaSampler(cond_p, state['precision'])
        cs, ts, acc, rate_log = [], [], [], []
        for _ in range(sweeps):                                                       # gibbs.py:136-151
            rw.state = state['coefficients']
            cond_c.precision = state['precision']
            before = rw._n_accepted_moves
            state['coefficients'] = rw.sample()
            acc.append(rw._n_accepted_moves > before)
            cond_p['coefficients'] = Value(state['coefficients'])
            state['precision'] = gs.sample()
            cs.append(np.array(state['coefficients'], copy=True))
            ts.append(state['precision'])
            rate_log.append(rw.acceptance_rate)
        path = os.path.join(OUT, 'ref_example_chain_%s.npz' % tag)
        np.savez_compressed(path, provenance=np.array(PROVENANCE), seed=np.int64(seed), stepsize=np.float64(stepsize),
                            xs=xs, ys=ys, coefficients=np.array(cs), precision=np.array(ts),
                            accepted=np.array(acc), acceptance_rate=np.array(rate_log),
                            gamma_shape=np.float64(gs._calculate_shape()),
                            last_draw_stats_field=np.array(rw.last_draw_stats['coefficients']._fields[0]))
        written.append(os.path.basename(path))
    print('wrote %d files to %s:\n  %s' % (len(written), OUT, '\n  '.join(written)))


if __name__ == '__main__':
    main()

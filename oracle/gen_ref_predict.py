"""
TEST INFRASTRUCTURE ONLY.  Writes tests/golden/ref_predict_*.npz: outputs of the REFERENCE'S OWN
posterior-predictive code (``binf/example/misc.py:3-16`` ``predict``,
``binf/example/plots.py:6-27`` ``plot_prediction_tube``), run in the build container.

PROVENANCE -- read this.  ``predict`` needs the absent third-party ``csb`` in its first statement
(``from csb.numeric import log_sum_exp``) and uses it in its last (``return np.exp(log_sum_exp(
integrands)) / len(samples)``); ``plot_prediction_tube`` calls ``predict`` and draws with matplotlib
(absent too).  Nothing is substituted for either.  What runs, unchanged, taken out of the
reference's syntax tree:

* ``predict``'s middle statement (``misc.py:7-14``: the ``if True:`` block that defines the
  integrand lambda and builds ``integrands``), executed in a namespace holding ``x``, ``y``,
  ``samples`` (data-only objects with a ``variables`` dict), ``polynomial``
  (``np.polynomial.polynomial.polyval``, ``example_script.py:21``) and ``np`` -> ``integrands``;
* ``plot_prediction_tube``'s statements at ``plots.py:8-9`` (the y grid), ``:12-16`` (``cdfs``,
  ``lower_tube_lims``, ``upper_tube_lims``) and the list comprehension of ``np.trapz`` inside the
  ``ax.plot`` call of ``:20-23`` -- with ``predicted_ys_probs`` (the one statement that calls
  ``predict``, ``:10-11``) handed in as DATA: densities computed by this repository's numpy
  restatement (oracle/ref_example.py).

So the fixtures pin the integrand and the tube's post-processing by reference output; the
definition of ``log_sum_exp`` and the final ``exp(...) / len(samples)`` stay **parity unpinned**.

Run (build container only):   python -m oracle.gen_ref_predict
"""
import ast
import os

import numpy as np

from oracle import ref_example

REF = '/root/reference/binf/example'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
PROVENANCE = ('outputs of the REFERENCE\'s own statements: binf/example/misc.py:7-14 (the integrands of '
              'predict) and binf/example/plots.py:8-9,12-16,20-23 (y grid, cdfs, 5%%/95%% limits, trapz '
              'mean of plot_prediction_tube), taken out of the syntax tree and executed unchanged; csb '
              'and matplotlib absent, nothing substituted: predict\'s import / return statements and '
              'the statement that calls predict (plots.py:10-11) are NOT run -- predicted_ys_probs is '
              'handed in as data from oracle/ref_example.py; numpy %s' % np.__version__)


def function(path, name):
    tree = ast.parse(open(path).read(), path)
    return next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)


def run(stmts, ns, where):
    mod = ast.Module(body=list(stmts), type_ignores=[])
    exec(compile(mod, where, 'exec'), ns)


class Sample(object):
    """data only: what ``x.variables`` of a BinfState holds"""

    def __init__(self, coefficients, precision):
        self.variables = dict(coefficients=coefficients, precision=precision)


def main():
    polynomial = np.polynomial.polynomial.polyval
    predict = function(os.path.join(REF, 'misc.py'), 'predict')
    assert [type(s).__name__ for s in predict.body] == ['ImportFrom', 'If', 'Return']
    tube = function(os.path.join(REF, 'plots.py'), 'plot_prediction_tube')
    kinds = [(type(s).__name__, s.lineno) for s in tube.body]
    assert kinds[:6] == [('ImportFrom', 6), ('Assign', 8), ('Assign', 10), ('Assign', 12), ('Assign', 13),
                         ('Assign', 15)], kinds
    trapz_call = tube.body[8]                                   # ax.plot(predict_space, [np.trapz(...) ...], ...)
    assert isinstance(trapz_call.value, ast.Call) and 'trapz' in ast.dump(trapz_call.value.args[1])
    trapz_expr = ast.Expression(body=trapz_call.value.args[1])

    for tag, S, K, nx, n_ys, seed, spread in [('s40_k4', 40, 4, 5, 60, 1, 0.1), ('s500_k4', 500, 4, 7, 150, 2, 0.1),
                                              ('s33_k9', 33, 9, 3, 40, 3, 0.01)]:
        rs = np.random.RandomState(seed)
        real = rs.standard_normal(K)
        coefficients = real + spread * rs.standard_normal((S, K))
        precisions = rs.gamma(20.0, 0.1, size=S)
        samples = [Sample(c, float(t)) for c, t in zip(coefficients, precisions)]
        predict_space = np.linspace(-1.5, 1.5, nx)
        centre = polynomial(predict_space, real)
        ys_from, ys_to = centre - 6.0, centre + 6.0
        # -- predict's integrands at a handful of points (misc.py:7-14) ---------------------------
        pts_x = rs.uniform(-1.5, 1.5, size=6)
        pts_y = polynomial(pts_x, real) + rs.standard_normal(6)
        integrands = []
        for x, y in zip(pts_x, pts_y):
            ns = dict(np=np, x=x, y=y, samples=samples, polynomial=polynomial)
            run([predict.body[1]], ns, 'misc.py:predict[:7-14]')
            integrands.append(ns['integrands'])
        # -- the tube's numbers (plots.py:8-9, 12-16, trapz of :20-23) ----------------------------
        ns = dict(np=np, samples=samples, polynomial=polynomial, predict_space=predict_space,
                  ys_from=ys_from, ys_to=ys_to, n_ys=n_ys)
        run([tube.body[1]], ns, 'plots.py:8-9')
        probs = ref_example.prediction_tube(coefficients, precisions, predict_space, ys_from, ys_to,
                                            n_ys)['probs']
        ns['predicted_ys_probs'] = probs                         # DATA, see the header
        run(tube.body[3:6], ns, 'plots.py:12-16')
        mean = np.array(eval(compile(ast.fix_missing_locations(trapz_expr), 'plots.py:20-23', 'eval'), ns))
        np.savez(os.path.join(OUT, 'ref_predict_%s.npz' % tag), provenance=PROVENANCE,
                 coefficients=coefficients, precisions=precisions, pts_x=pts_x, pts_y=pts_y,
                 integrands=np.array(integrands), predict_space=predict_space, ys_from=ys_from,
                 ys_to=ys_to, n_ys=n_ys, predicted_ys=ns['predicted_ys'], probs_in=probs,
                 cdfs=ns['cdfs'], lower=ns['lower_tube_lims'], upper=ns['upper_tube_lims'],
                 prediction=mean)
        print('wrote ref_predict_%s.npz' % tag)


if __name__ == '__main__':
    main()

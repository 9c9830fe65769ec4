"""Factories and the posterior-predictive helpers of the example application (mirror of
reference ``binf/example/misc.py``; of ``binf/example/plots.py`` the NUMBERS that
``plot_prediction_tube`` draws -- the drawing itself is out of scope).

The reference's ``predict`` loops over the samples in Python for ONE (x, y) point, and
``plot_prediction_tube`` calls it for every point of a grid (``plots.py:10-11``: 100 x 150
points x 500 samples).  Here the samples stay in HBM as ``[S x K]`` / ``[S]`` tensors and
the whole grid is one launch (``binf_predictive_density_f64``, ``csrc/predict.hip``)."""
import collections
import math

import numpy as np
import torch

from binf_amd import _native


def _samples_as_tensors(samples):
    """``samples``: what the reference's sampling loop collects -- a sequence of ``BinfState``
    (``example_script.py:32-34``; each holding ``coefficients`` ``[K]`` or, chain-batched,
    ``[C x K]``, and ``precision`` as a number or a ``[C]`` tensor) -- or, already stacked, a pair
    ``(coefficients [S x K], precision [S])`` of device tensors (a gathered ``SampleStore``
    block).  Returns that pair; every chain of every state counts as one sample."""
    if isinstance(samples, tuple) and len(samples) == 2 and isinstance(samples[0], torch.Tensor):
        coefficients, precision = samples
        coefficients = coefficients.reshape(-1, coefficients.shape[-1])
    else:
        if len(samples) == 0:
            raise ValueError('predict: no samples (the reference takes max() of an empty array)')
        cs, ps = [], []
        for s in samples:
            v = s.variables
            c = v['coefficients']
            _native.require_device(c, 'coefficients')
            c = c.reshape(-1, c.shape[-1])
            p = v['precision']
            if not isinstance(p, torch.Tensor):
                p = torch.full((c.shape[0],), float(p), dtype=torch.float64, device=c.device)
            cs.append(c)
            ps.append(p.reshape(-1).expand(c.shape[0]) if p.numel() == 1 else p.reshape(-1))
        coefficients, precision = torch.cat(cs, 0), torch.cat(ps, 0)
    precision = precision.reshape(-1)
    if precision.shape[0] != coefficients.shape[0]:
        raise ValueError('predict: %d coefficient vectors but %d precisions'
                         % (coefficients.shape[0], precision.shape[0]))
    return coefficients.contiguous(), precision.contiguous()


def _on_device(a, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float64).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64))).to(device)


def predict_grid(xs, ys, samples, polynomial):
    """``predict(xs[i], ys[i, j], samples, polynomial)`` (``binf/example/misc.py:3-16``) for every
    point of the grid at once: ``xs`` ``[nx]``, ``ys`` ``[nx x ny]`` -> device tensor ``[nx x ny]``.
    ``polynomial``: ``np.polynomial.polynomial.polyval`` (``example_script.py:21``) runs as the
    Horner kernel; any other callable is applied as given,
    ``polynomial(xs, coefficients [S x K]) -> [S x nx]`` device tensor."""
    from binf_amd.example.likelihood import POLYVAL
    coefficients, precision = _samples_as_tensors(samples)
    dev = coefficients.device
    xs_d, ys_d = _on_device(xs, dev).reshape(-1), _on_device(ys, dev)
    if ys_d.dim() != 2 or ys_d.shape[0] != xs_d.shape[0]:
        raise ValueError('predict_grid: ys must be [len(xs) x ny]')
    if polynomial is POLYVAL:
        mock = _native.poly_forward(coefficients, xs_d)
    else:
        mock = polynomial(xs_d, coefficients)
        _native.require_device(mock, 'polynomial(xs, coefficients)')
        mock = mock.reshape(coefficients.shape[0], xs_d.shape[0]).contiguous()
    return _native.predictive_density(mock, precision, ys_d, 0.5 * math.log(2.0 * math.pi))


def predict(x, y, samples, polynomial):
    """Posterior-predictive density of ``y`` at ``x`` (``binf/example/misc.py:3-16``): a float."""
    return float(predict_grid([float(x)], [[float(y)]], samples, polynomial)[0, 0])


PredictionTube = collections.namedtuple(
    'PredictionTube', 'predicted_ys probs cdfs lower upper prediction')


def prediction_tube(samples, polynomial, predict_space, ys_from, ys_to, n_ys):
    """The numbers ``plot_prediction_tube`` draws (``binf/example/plots.py:8-27``): per x of
    ``predict_space`` a y grid from ``ys_from[i]`` to ``ys_to[i]``, the predictive density on it
    (one launch for the whole grid), its running integral, the 5 % / 95 % limits and the
    trapezoid mean.  Host arrays (presentation data; the ``[nx x n_ys]`` densities are copied back
    once).  Like the reference it raises ``IndexError`` when the y range misses a limit."""
    predict_space = np.asarray(predict_space, dtype=np.float64)
    predicted_ys = np.array([np.linspace(ys_from[i], ys_to[i], n_ys)
                             for i, _ in enumerate(predict_space)])
    probs = predict_grid(predict_space, predicted_ys, samples, polynomial).cpu().numpy()
    cdfs = np.cumsum(probs * (predicted_ys[:, 1] - predicted_ys[:, 0])[:, None], 1)
    lower = np.array([predicted_ys[i][np.where(cdfs[i] < 0.05)[0][-1]]
                      for i in range(len(predict_space))])
    upper = np.array([predicted_ys[i][np.where(cdfs[i] > 0.95)[0][0]]
                      for i in range(len(predict_space))])
    trapezoid = getattr(np, 'trapezoid', None) or np.trapz
    prediction = np.array([trapezoid(predicted_ys[i] * probs[i], predicted_ys[i])
                           for i in range(len(predict_space))])
    return PredictionTube(predicted_ys, probs, cdfs, lower, upper, prediction)


def get_MAP(samples, log_probs):
    best = samples[int(np.argmax(log_probs))]
    return best.variables['coefficients'], best.variables['precision']


def make_posterior(xses, ys, polynomial):
    from binf_amd.example.likelihood import make_likelihood
    from binf_amd.example.priors import make_priors
    from binf_amd.pdf.posteriors import Posterior
    L = make_likelihood(xses, ys, polynomial)
    return Posterior({L.name: L}, make_priors())

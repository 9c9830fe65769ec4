"""A larger statistical look at the draws the sampling kernels generate for
themselves (xoshiro128++ streams + 1024-layer ziggurat, csrc/xoshiro.hpp), on the dump
of the stream at C2's shape: 2.7e8 normals, 2.6e5 uniforms.  Not part of the suite
(needs scipy and ~20 s of GPU + host time); figures recorded in profiles/."""
import json, os, sys
import numpy as np, torch
from scipy import stats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
dev = torch.device('cuda:0')
C, D, n = 4096, 1024, 64
p0, u = _native.hmc_gauss_rng_draws(n, C, D, 20260104, 3, dev)
N = p0.numel()
out = {'n_normals': N, 'n_uniforms': u.numel()}
# moments on the device (float64)
m = [float((p0 ** k).mean()) for k in (1, 2, 3, 4, 6, 8)]
out['moments_1_2_3_4_6_8'] = m
out['moment_z'] = {  # (estimate - expectation) / standard error
    'mean': m[0] / np.sqrt(1.0 / N), 'var': (m[1] - 1) / np.sqrt(2.0 / N),
    'skew': m[2] / np.sqrt(15.0 / N), 'kurt4': (m[3] - 3) / np.sqrt(96.0 / N),
    'm6': (m[4] - 15) / np.sqrt((10395 - 225) / N), 'm8': (m[5] - 105) / np.sqrt((2027025 - 11025) / N)}
# chi-square of a 400-bin histogram with equal-probability bins
edges = torch.from_numpy(stats.norm.ppf(np.linspace(0, 1, 401)[1:-1])).to(dev)
counts = torch.bincount(torch.bucketize(p0.reshape(-1), edges), minlength=400).double().cpu().numpy()
chi2 = float(((counts - N / 400.0) ** 2 / (N / 400.0)).sum())
out['chi2_400_bins'] = chi2
out['chi2_p_value'] = float(stats.chi2.sf(chi2, 399))
# tails
for t in (3.0, 4.038849846109505, 4.5, 5.0, 5.5):
    got = int((p0.abs() > t).sum())
    want = 2 * stats.norm.sf(t) * N
    out['tail_%.3g' % t] = {'count': got, 'expected': want, 'z': (got - want) / np.sqrt(want)}
# correlations: (a) successive draws of one lane (t -> t+1: elements 8 apart), lags 1..4;
# (b) neighbouring lanes; (c) same element, next transition; (d) same lane slot, next chain
x = p0
def corr(a, b):
    a = a.reshape(-1); b = b.reshape(-1)
    return float(((a - a.mean()) * (b - b.mean())).mean() / (a.std() * b.std()))
se = 1.0 / np.sqrt(N)
out['corr_se'] = se
out['corr_same_lane_lag'] = [corr(x[..., :-8 * k], x[..., 8 * k:]) / se for k in (1, 2, 3, 4)]
out['corr_neighbour_lane'] = corr(x[..., :-1], x[..., 1:]) / se
out['corr_next_transition'] = corr(x[:-1], x[1:]) / se
out['corr_next_chain'] = corr(x[:, :-1], x[:, 1:]) / se
# squares (dependence beyond linear) for successive draws of a lane
out['corr_squares_same_lane_lag1'] = corr(x[..., :-8] ** 2, x[..., 8:] ** 2) / se
# uniforms
uu = u.reshape(-1).cpu().numpy()
out['uniform_ks_p'] = float(stats.kstest(uu, 'uniform').pvalue)
out['uniform_mean_z'] = float((uu.mean() - 0.5) / np.sqrt(1 / 12 / uu.size))
out['normal_ks_p_1e6_subsample'] = float(stats.kstest(p0.reshape(-1)[::257][:1000000].cpu().numpy(), 'norm').pvalue)
print(json.dumps(out, indent=1))

"""Does the chi^2 memo of the pair-distance log-prob hit for E_before of the next sample()?
Prints, per sample(), how many chains were skipped in each of the two log-prob calls."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
from binf_amd.example import distance as DM
from binf_amd.example.distance import make_distance_likelihood
from binf_amd.pdf import IsotropicGaussian
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')
n, L = 256, 20
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rs = np.random.RandomState(0)
truth = rs.standard_normal((n, 3)) * 2.0
I_, J_ = np.triu_indices(n, 1)
ys = np.abs(np.sqrt(np.sum((truth[I_] - truth[J_]) ** 2, axis=1)) + 0.05 * rs.standard_normal(n * (n - 1) // 2))
x = torch.from_numpy(truth.reshape(-1)[None, :] + 0.1 * rs.standard_normal((C, 3 * n))).to(dev)
lik = make_distance_likelihood(ys, n)
prior = IsotropicGaussian(0.05, 0.0, name='coordinates_prior', variable_name='coordinates')
cond = Posterior({lik.name: lik}, {prior.name: prior}).conditional_factory(precision=4.0)
s = HMCSampler(cond, x, 0.002, L, variable_name='coordinates', rng=DeviceRNG(0, dev))
orig = _native.pairdist_gauss_logp_memo
log = []
def spy(x2, I, J, ys_, prec, memo):
    out = orig(x2, I, J, ys_, prec, memo)
    log.append(int(memo[2][0].sum()))
    return out
_native.pairdist_gauss_logp_memo = spy
DM._native.pairdist_gauss_logp_memo = spy
for i in range(5):
    log.clear()
    s.sample()
    print('sample', i, 'skipped chains per log-prob call:', log, 'accepted', int(s.last_move_accepted.sum()))

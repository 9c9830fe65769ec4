"""Host-side cost of one HMCSampler.sample() on the C5 posterior (cProfile, top
entries by own time), next to its wall time (development aid)."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.example.distance import make_distance_likelihood
from binf_amd.pdf import IsotropicGaussian
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0'); n, C, L = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 20
rs = np.random.RandomState(0)
truth = rs.standard_normal((n, 3)) * 2.0
I, J = np.triu_indices(n, 1)
ys = np.abs(np.sqrt(((truth[I] - truth[J]) ** 2).sum(1)) + 0.05 * rs.standard_normal(len(I)))
x = torch.from_numpy(truth.reshape(-1)[None, :] + 0.1 * rs.standard_normal((C, 3 * n))).to(dev)
lik = make_distance_likelihood(ys, n)
prior = IsotropicGaussian(0.05, 0.0, name='coordinates_prior', variable_name='coordinates')
cond = Posterior({lik.name: lik}, {prior.name: prior}).conditional_factory(precision=4.0)
s = HMCSampler(cond, x, 0.002, L, variable_name='coordinates', rng=DeviceRNG(0, dev))
for _ in range(20): s.sample()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200): s.sample()
t_issue = (time.perf_counter() - t) / 200
torch.cuda.synchronize()
t_wall = (time.perf_counter() - t) / 200
print('sample(): %.1f us to issue, %.1f us wall' % (t_issue * 1e6, t_wall * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): s.sample()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(22)

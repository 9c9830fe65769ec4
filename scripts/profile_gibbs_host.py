"""Host-side cost of one Gibbs sweep on the example model (cProfile; development aid)."""
import cProfile, pstats, os, sys, io
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.example.misc import make_posterior
from binf_amd.example.samplers import make_hmc_sampler
from binf_amd.samplers import BinfState
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0'); C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
polynomial = np.polynomial.polynomial.polyval
xses = np.linspace(-2, 2, 20)
ys = np.random.RandomState(0).normal(loc=polynomial(xses, np.array([2., -4., 1., 1.5])), scale=1 / np.sqrt(2.5))
start = BinfState(dict(coefficients=torch.ones((C, 4), dtype=torch.float64, device=dev),
                       precision=torch.ones(C, dtype=torch.float64, device=dev)))
rng = DeviceRNG(0, dev)
gips = make_hmc_sampler(make_posterior(xses, ys, polynomial), 0.02, 50, start, rng=rng, gamma=rng.gamma)
for _ in range(50): gips.sample()
torch.cuda.synchronize()
import time
t = time.perf_counter()
for _ in range(500): gips.sample()
torch.cuda.synchronize()
print('%.1f us per sweep' % ((time.perf_counter() - t) / 500 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(300): gips.sample()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:4500])

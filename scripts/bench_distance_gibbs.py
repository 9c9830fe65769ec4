"""Gibbs-within-HMC on the restraint posterior (C5's model, one precision per chain): time per
sweep next to the plain HMC sample() of scripts/bench_distance.py.
  python scripts/bench_distance_gibbs.py [chains]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd.example.distance import make_distance_likelihood, make_restraint_gibbs_sampler
from binf_amd.example.priors import GammaPrior
from binf_amd.pdf import IsotropicGaussian
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers import BinfState
from binf_amd.samplers.rng import DeviceRNG

dev = torch.device('cuda:0')
n, L = 256, 20
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rs = np.random.RandomState(0)
truth = rs.standard_normal((n, 3)) * 2.0
I_, J_ = np.triu_indices(n, 1)
ys = np.abs(np.sqrt(np.sum((truth[I_] - truth[J_]) ** 2, axis=1)) + 0.5 * rs.standard_normal(n * (n - 1) // 2))
lik = make_distance_likelihood(ys, n)
post = Posterior({lik.name: lik},
                 {'coordinates_prior': IsotropicGaussian(0.05, 0.0, name='coordinates_prior', variable_name='coordinates'),
                  'precision_prior': GammaPrior(1.0, 0.2)})
rng = DeviceRNG(0, dev)
start = BinfState({'coordinates': torch.from_numpy(truth.reshape(-1)[None, :] + 0.1 * rs.standard_normal((C, 3 * n))).to(dev),
                   'precision': torch.full((C,), 4.0, dtype=torch.float64, device=dev)})
gips = make_restraint_gibbs_sampler(post, 0.002, L, start, rng=rng)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    gips.sample(); torch.cuda.synchronize()
k = 40
t = time.perf_counter()
for _ in range(k):
    gips.sample()
host = (time.perf_counter() - t) / k
torch.cuda.synchronize()
tot = (time.perf_counter() - t) / k
st = gips.sample()
print(json.dumps({'config': {'chains': C, 'beads': n, 'L': L}, 'gibbs_sweep_ms': tot * 1e3, 'host_issue_ms': host * 1e3,
                  'chain_leapfrog_steps_per_s': C * L / tot,
                  'acceptance': float(gips.subsamplers['coordinates'].acceptance_rate.mean()),
                  'precision_mean': float(st.variables['precision'].mean())}))
if os.environ.get('BINF_PROFILE_HOST'):
    import cProfile, pstats, io
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        gips.sample()
    pr.disable()
    torch.cuda.synchronize()
    sio = io.StringIO()
    pstats.Stats(pr, stream=sio).sort_stats('cumtime').print_stats(45)
    print(sio.getvalue()[:7000])

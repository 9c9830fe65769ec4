"""Per-step (generic) tier on a user-style PDF written with torch ops, at C2's
shape: what a plug-in posterior without a fused kernel costs (development aid)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')


class TorchGaussian(object):
    def __init__(self, k, x0):
        self.k, self.x0 = k, x0

    def log_prob(self, x):
        return (-0.5 * self.k) * _native.row_sum(x, _native.ROW_SUMSQ_SHIFT, shift=self.x0)

    def gradient(self, x):
        return self.k * (x - self.x0)


out = {}
for C, D, L, graph in ((4096, 1024, 20, False), (256, 768, 20, False), (256, 768, 20, True),
                       (2048, 64, 10, False), (2048, 64, 10, True), (64, 20000, 10, False)):
    q0 = torch.randn((C, D), dtype=torch.float64, device=dev)
    s = HMCSampler(TorchGaussian(1.0, 0.0), q0, 0.05, L, variable_name='x', rng=DeviceRNG(0, dev), graph=graph)
    for _ in range(3): s.sample()
    torch.cuda.synchronize(); t = time.perf_counter()
    K = 10
    for _ in range(K): s.sample()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / K
    bytes_step = 32.0 * D * C          # unfused kick + drift traffic per leapfrog step
    out['%dx%d L=%d%s' % (C, D, L, ' graph' if graph else '')] = {'ms_per_sample': dt * 1e3, 'us_per_leapfrog_step': dt / L * 1e6,
                                     'chain_steps_per_s': C * L / dt,
                                     'kick_drift_GBps_if_alone': bytes_step * L / dt / 1e9}

# The Likelihood plug-in surface with a user's forward model (no fused kernel of its own):
# forward pass and error-model gradient are the user's / the example's classes, the chain rule
# is binf_jacobian_contract_f64, the Posterior's sums binf_sum_terms_f64 (likelihoods.py:141-155)
from binf_amd.example.likelihood import POLYVAL, ForwardModel, GaussianErrorModel
from binf_amd.example.priors import GammaPrior, GaussianPrior
from binf_amd.pdf.likelihoods import Likelihood
from binf_amd.pdf.posteriors import Posterior


class PlainPolynomial(ForwardModel):
    def _evaluate(self, coefficients):           # an override: no fused kernel may be assumed
        return ForwardModel._evaluate(self, coefficients)


for C, K, N, L, graph in ((4096, 8, 512, 20, False), (4096, 8, 512, 20, True), (8192, 33, 16384, 20, False)):
    rs = np.random.RandomState(0)
    xs = np.linspace(-1, 1, N)
    ys = POLYVAL(xs, rs.standard_normal(K)) + rs.standard_normal(N) / np.sqrt(2.5)
    lik = Likelihood('points', PlainPolynomial(xs, POLYVAL), GaussianErrorModel(ys))
    post = Posterior({lik.name: lik}, {'precision_prior': GammaPrior(1.0, 0.2),
                                       'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K) * 5)})
    cond = post.conditional_factory(precision=torch.full((C,), 2.5, dtype=torch.float64, device=dev))
    assert cond.native_hmc_spec('coefficients') is None
    q0 = torch.from_numpy(rs.standard_normal((C, K)) * 0.1).to(dev)
    s = HMCSampler(cond, q0, 1e-4, L, variable_name='coefficients', rng=DeviceRNG(0, dev, fused=False), graph=graph)
    for _ in range(3): s.sample()
    torch.cuda.synchronize(); t = time.perf_counter()
    R = 10 if N < 10000 else 4
    for _ in range(R): s.sample()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / R
    out['plug-in model %d chains, K=%d, N=%d, L=%d%s' % (C, K, N, L, ' graph' if graph else '')] = {
        'ms_per_sample': dt * 1e3, 'us_per_leapfrog_step': dt / L * 1e6, 'chain_steps_per_s': C * L / dt}
print(json.dumps(out))

"""Randomised differential soak of the round-3 kernels (development aid, run on the GPU):
* GibbsSampler.sample_n (binf_gibbs_poly_sample_n_f64) == the loop of gips.sample(),
  bit for bit: random K <= 16, N <= 1024 (any pairwise tree the kernel covers), chains,
  sweeps, thinning, HMC / RWMC, exact / fma, adaption windows, one / two generators,
  sharded launches with chain offsets;
* binf_jacobian_contract_f64 vs numpy within 1e-10 sum|J||r|, shared and per-chain,
  and batch independence; binf_sum_terms_f64 vs the sequential sum, bit for bit.
  python tests/soak/fuzz_gibbs_n.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from binf_amd import _native
from binf_amd.example.likelihood import POLYVAL, ForwardModel, GaussianErrorModel
from binf_amd.example.priors import GammaPrior, GaussianPrior
from binf_amd.example.samplers import make_hmc_sampler, make_sampler
from binf_amd.pdf.likelihoods import Likelihood
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers import BinfState
from binf_amd.samplers.rng import DeviceRNG

dev = torch.device('cuda:0')
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
bad = 0
t0 = time.time()
fused_launches = [0]
_orig_launch = _native.gibbs_poly_sample_n


def _counting(*a, **k):
    fused_launches[0] += 1
    return _orig_launch(*a, **k)


_native.gibbs_poly_sample_n = _counting


def report(what, **kw):
    global bad
    bad += 1
    print('MISMATCH', what, kw, flush=True)


def build(move, K, N, C, seed, rng, L, dt, start, **kw):
    r = np.random.RandomState(seed)
    xs = np.linspace(-1.5, 1.5, N) if N > 1 else np.array([0.3])
    ys = POLYVAL(xs, r.standard_normal(K)) + r.standard_normal(N) / np.sqrt(2.5)
    lik = Likelihood('points', ForwardModel(xs, POLYVAL), GaussianErrorModel(ys))
    post = Posterior({lik.name: lik}, {'precision_prior': GammaPrior(1.0, 0.2),
                                       'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K) * 5)})
    st = BinfState(dict(coefficients=start[0].clone(), precision=start[1].clone()))
    if move == 'hmc':
        return make_hmc_sampler(post, dt, L, st, rng=rng, **kw)
    return make_sampler(post, 0.05, st, rng=rng)


for case in range(n_cases):
    K = int(rs.randint(1, 17))
    N = int(rs.choice([2, 3, 7, 8, 9, 20, 37, 64, 127, 128, 129, 200, 256, 300, 511, 513, 777, 920, 1024]))
    if _native.pairwise_tree_height(N) > 3:
        N = 920
    C = int(rs.choice([1, 2, 7, 8, 9, 33, 100, 257]))
    n = int(rs.randint(1, 9))
    thin = int(rs.randint(1, n + 1))
    move = 'hmc' if rs.rand() < 0.6 else 'rwmc'
    L = int(rs.randint(1, 12))
    dt = 0.02 / (K * np.sqrt(N / 20.0))
    seed = int(rs.randint(0, 2 ** 31))
    kw = {}
    if move == 'hmc':
        kw['mode'] = 'fma' if rs.rand() < 0.3 else 'exact'
        if rs.rand() < 0.4:
            kw['timestep_adaption_limit'] = int(rs.randint(2, 8))
    r2 = np.random.RandomState(seed + 1)
    start = (t(r2.standard_normal((C, K))), t(1.0 + r2.uniform(size=C)))
    a = build(move, K, N, C, seed, DeviceRNG(seed % 1000, dev), L, dt, start, **kw)
    b = build(move, K, N, C, seed, DeviceRNG(seed % 1000, dev), L, dt, start, **kw)
    cs, ts = [], []
    a.fused_sweep = False                       # the per-variable sweep, kernel by kernel
    for _ in range(n):
        s = a.sample()
        cs.append(s.variables['coefficients'].clone())
        ts.append(s.variables['precision'].clone())
    cs, ts = torch.stack(cs), torch.stack(ts)
    rec = b.sample_n(n, thin=thin)
    ok = torch.equal(b.state.variables['coefficients'], cs[-1]) and \
        torch.equal(b.state.variables['precision'], ts[-1])
    if n // thin > 0:
        ok = ok and torch.equal(rec['coefficients'], cs[thin - 1::thin][:n // thin]) and \
            torch.equal(rec['precision'], ts[thin - 1::thin][:n // thin])
    sa, sb = a.subsamplers['coefficients'], b.subsamplers['coefficients']
    if move == 'hmc':
        ok = ok and torch.equal(sa.n_accepted, sb.n_accepted)
        if kw.get('timestep_adaption_limit'):
            ok = ok and torch.equal(torch.as_tensor(sa.timestep), torch.as_tensor(sb.timestep))
    if not ok or not torch.isfinite(cs).all():
        report('gibbs sample_n', move=move, K=K, N=N, C=C, n=n, thin=thin, L=L, kw=kw, seed=seed)
    # a shard of it with the chain offset
    if C >= 2:
        lo = int(rs.randint(0, C - 1))
        hi = int(rs.randint(lo + 1, C + 1))
        c = build(move, K, N, hi - lo, seed, DeviceRNG(seed % 1000, dev, chain_offset=lo), L, dt,
                  (start[0][lo:hi], start[1][lo:hi]), **kw)
        c.sample_n(n, record=False)
        if not (torch.equal(c.state.variables['coefficients'], cs[-1][lo:hi]) and
                torch.equal(c.state.variables['precision'], ts[-1][lo:hi])):
            report('gibbs shard', move=move, K=K, N=N, C=C, lo=lo, hi=hi, seed=seed)

    # ---- contraction --------------------------------------------------------------
    Kc = int(rs.randint(1, 140))
    Nc = int(rs.choice([1, 2, 63, 64, 65, 100, 500, 1000, 4097]))
    Cc = int(rs.choice([1, 3, 16, 17, 40]))
    batched = rs.rand() < 0.5
    J = rs.standard_normal((Cc, Kc, Nc) if batched else (Kc, Nc)) * 10.0 ** rs.randint(-3, 4)
    r = rs.standard_normal((Cc, Nc))
    got = _native.jacobian_contract(t(J), t(r)).cpu().numpy()
    want = np.einsum('ckn,cn->ck', J, r) if batched else r.dot(J.T)
    bar = 1e-10 * (np.einsum('ckn,cn->ck', np.abs(J), np.abs(r)) if batched else np.abs(r).dot(np.abs(J).T))
    if not np.all(np.abs(got - want) <= bar):
        report('contraction', K=Kc, N=Nc, C=Cc, batched=batched)
    lo = int(rs.randint(0, Cc))
    part = _native.jacobian_contract(t(J[lo:] if batched else J), t(r[lo:])).cpu().numpy()
    if not np.array_equal(part, got[lo:]):
        report('contraction batch independence', K=Kc, N=Nc, C=Cc, batched=batched, lo=lo)
    # ---- term sum ---------------------------------------------------------------------
    T = int(rs.randint(2, 17))
    m = int(rs.choice([1, 5, 1000]))
    vals = [rs.standard_normal(m) * 10.0 ** rs.randint(-6, 7) if rs.rand() < 0.8 else float(rs.standard_normal())
            for _ in range(T)]
    if not any(isinstance(v, np.ndarray) for v in vals):
        vals[0] = rs.standard_normal(m)
    got = _native.sum_terms([t(v) if isinstance(v, np.ndarray) else v for v in vals]).cpu().numpy()
    want = vals[0] if isinstance(vals[0], np.ndarray) else np.full(m, vals[0])
    for v in vals[1:]:
        want = want + v
    if not np.array_equal(got, want):
        report('sum_terms', T=T, m=m)
    if case % 50 == 49:
        print('%d cases, %d mismatches, %.0f s' % (case + 1, bad, time.time() - t0), flush=True)

print('fuzz_gibbs_n: %d cases, %d mismatches, %d fused multi-sweep launches, %.0f s'
      % (n_cases, bad, fused_launches[0], time.time() - t0))
sys.exit(1 if bad else 0)

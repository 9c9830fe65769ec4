"""Randomised differential tests of the model kernels (development aid / soak):
* pair distances bit for bit, all-pairs force to 1e-10 vs oracle/ref_distance.py;
  fused leapfrog vs per-step tier bit for bit, results independent of the batch
  size (1-lane / 4-lane force variants);
* MFMA polynomial gradient vs the numpy chain rule (1e-10);
* fused small-data polynomial transition vs the per-step tier (E_before bit
  for bit, state to 1e-10, same flags);
* Gibbs-within-HMC sweeps (fused transition + conjugate precision update)
  through the class stack vs the single-chain numpy restatement.
  python tests/soak/fuzz_models.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from binf_amd import _native
from binf_amd.example.distance import make_distance_likelihood
from binf_amd.example.likelihood import POLYVAL, ForwardModel, GaussianErrorModel
from binf_amd.example.priors import GammaPrior, GaussianPrior
from binf_amd.pdf import IsotropicGaussian
from binf_amd.pdf.likelihoods import Likelihood
from binf_amd.pdf.posteriors import Posterior
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.example.samplers import make_hmc_sampler
from binf_amd.samplers import BinfState
from oracle import ref_distance as RD
from oracle import ref_example as RE
from oracle import ref_numpy as R

dev = torch.device('cuda:0')
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
bad = 0
t0 = time.time()


def report(what, **kw):
    global bad
    bad += 1
    print('MISMATCH', what, kw, flush=True)


for case in range(n_cases):
    # ---- pair distances ---------------------------------------------------
    n = int(rs.choice([2, 3, 5, 17, 64, 100, 255, 256, 257, 320, 400, 512, 513, 700, 1000, 1024]))   # 257..1024: the ring kernels
    C = int(rs.choice([1, 3, 9]))
    truth = rs.standard_normal((n, 3)) * 2.0
    ys = np.abs(RD.forward(truth.reshape(-1), n) + 0.05 * rs.standard_normal(n * (n - 1) // 2))
    x = truth.reshape(-1)[None, :] + 0.3 * rs.standard_normal((C, 3 * n))
    lik = make_distance_likelihood(ys, n)
    tau = rs.uniform(0.5, 4.0, size=C)
    got = lik.forward_model(coordinates=t(x)).cpu().numpy()
    if not np.array_equal(got, np.stack([RD.forward(x[c], n) for c in range(C)])):
        report('pairdist forward', n=n, C=C)
    g = lik.gradient(coordinates=t(x), precision=t(tau)).cpu().numpy()
    for c in range(C):
        w = RD.gradient(x[c], ys, tau[c], n)
        # scale of the sum: tau * sum_j |x_i - x_j| (each pair's weight (d - y)/d is O(1) and is
        # evaluated as 1 - y * rsqrt(d^2): absolute, not relative, accuracy when d ~ y)
        xc = x[c].reshape(n, 3)
        scale = tau[c] * np.abs(xc[:, None, :] - xc[None, :, :]).sum(axis=1).max()
        if np.abs(g[c] - w).max() > 1e-10 * max(np.abs(w).max(), scale):
            report('pairdist force', n=n, C=C, c=c)
    # batch-size independence across the 1-lane / 4-lane switch (C < 1024, n <= 512)
    if n <= 300 and case % 4 == 0:
        big = np.concatenate([x, x[:1].repeat(1100, 0) + 0.1 * rs.standard_normal((1100, 3 * n))])
        gb = lik.gradient(coordinates=t(big), precision=2.0).cpu().numpy()
        gs = lik.gradient(coordinates=t(x), precision=2.0).cpu().numpy()
        if not np.array_equal(gb[:C], gs):
            report('pairdist batch independence', n=n, C=C)
    # chi^2 with two chains per workgroup (>= 2048 chains) vs one: the same bits, and numpy's
    if n <= 300 and case % 8 == 1:
        extra = 2047 + int(rs.randint(3))
        big = np.concatenate([x, x[:1].repeat(extra, 0) + 0.1 * rs.standard_normal((extra, 3 * n))])
        lb = lik.log_prob(coordinates=t(big), precision=2.0).cpu().numpy()
        ls = lik.log_prob(coordinates=t(x), precision=2.0).cpu().numpy()
        if not np.array_equal(lb[:C], ls):
            report('pairdist log-prob, chains per workgroup', n=n, C=C)
        c = int(rs.randint(len(big)))
        if lb[c] != RD.log_prob(big[c], ys, 2.0, n):
            report('pairdist log-prob vs numpy', n=n, c=c)
    # fused leapfrog vs per-step
    with_prior = bool(rs.randint(2))
    priors = {}
    if with_prior:
        name = str(rs.choice(['a_prior', 'z_prior']))        # before / after the likelihood
        priors[name] = IsotropicGaussian(0.05, 0.1, name=name, variable_name='coordinates')
    cond = Posterior({lik.name: lik}, priors).conditional_factory(precision=t(tau))
    L = int(rs.randint(1, 5))
    p0, u = rs.standard_normal((C, 3 * n)), rs.uniform(size=C)
    outs = []
    for fused in (True, False):
        s = HMCSampler(cond, t(x), 0.002, L, variable_name='coordinates',
                       mode=str(rs.choice(['exact', 'fma'])) if False else 'exact')
        s.fused_leapfrog = fused
        o = s.sample(p0=t(p0), u=t(u))
        outs.append((o.cpu().numpy(), s.last_e_after.cpu().numpy()))
    if not (np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])):
        report('pairdist fused leapfrog', n=n, C=C, L=L, prior=with_prior)
    # one-launch energy (+ memo, packed targets) vs the per-step energy: three consecutive
    # transitions with steps long enough to reject, adapted step sizes
    if n <= 2048:
        runs = []
        step = float(rs.choice([0.002, 0.8, 1.6, 2.6])) / np.sqrt(3.0 * n) if rs.randint(2) else 0.002
        draws = [(t(rs.standard_normal((C, 3 * n))), t(rs.uniform(size=C))) for _ in range(3)]
        for fused in (True, False):
            s = HMCSampler(cond, t(x), step, 4, timestep_adaption_limit=3, variable_name='coordinates',
                           record_energies=True)
            s.fused_energy = fused
            o = [s.sample(p0=a, u=b).clone() for a, b in draws]
            runs.append((torch.stack(o), s.last_e_before.clone(), s.last_e_after.clone(), s.n_accepted.clone(),
                         s.timestep.clone()))
        if not all(torch.equal(a, b) for a, b in zip(runs[0], runs[1])):
            report('pairdist one-launch energy', n=n, C=C, prior=with_prior, step=step)
        if 32 <= n <= 256:
            em = lik.error_model
            ym, pk = em.ymat_device(dev), em.ypacked_device(dev)
            if not torch.equal(_native.pairdist_gauss_grad(t(x), ym, 2.0, packed=pk),
                               _native.pairdist_gauss_grad(t(x), ym, 2.0)):
                report('pairdist packed targets', n=n, C=C)

    # ---- MFMA gradient ------------------------------------------------------
    K = int(rs.randint(1, 65))
    N = int(rs.choice([1, 15, 16, 17, 32, 100, 160, 1000, 1024, 4096, 4099, 16384]))   # incl. whole-tile sets (N % 16 == 0): the trimmed MFMA kernel
    Cg = int(rs.choice([1, 15, 16, 17, 63, 64, 65, 130, 2100]))
    if Cg * N > 4e6:
        Cg = 17
    xs = np.linspace(-1, 1, N)
    ysp = rs.standard_normal(N)
    th = rs.standard_normal((Cg, K)) * 0.5
    taug = rs.uniform(0.5, 3.0, size=Cg)
    A = np.vstack([xs ** i for i in range(K)])
    got = _native.poly_gauss_grad(t(th), t(A), t(ysp), t(taug)).cpu().numpy()
    want = ((th @ A - ysp) * taug[:, None]) @ A.T
    if np.abs(got - want).max() > 1e-10 * max(np.abs(want).max(), 1e-300):
        report('mfma gradient', K=K, N=N, C=Cg, err=float(np.abs(got - want).max() / np.abs(want).max()))

    # ---- fused small polynomial transition -----------------------------------
    K = int(rs.randint(1, 17))
    # one lane per chain up to 128 data points, one wave per chain up to 1024
    N = int([rs.randint(1, 129), rs.randint(129, 921), 8 * rs.randint(17, 129)][rs.randint(3)])
    Cs = int(rs.choice([1, 5, 64, 65, 200]))
    xs = np.linspace(-1.5, 1.5, N)
    ysp = R.polyval(xs, rs.standard_normal(K)) + 0.5 * rs.standard_normal(N)
    lik_name = str(rs.choice(['a_points', 'points']))
    lp = Likelihood(lik_name, ForwardModel(xs, POLYVAL), GaussianErrorModel(ysp))
    priors = {'precision_prior': GammaPrior(1.0, 0.2)}
    if rs.randint(2):
        priors['coefficients_prior'] = GaussianPrior(rs.standard_normal(K), rs.uniform(1, 5, K))
    tau = t(rs.uniform(0.5, 2.0, size=Cs)) if rs.randint(2) else float(rs.uniform(0.5, 2.0))
    cond = Posterior({lp.name: lp}, priors).conditional_factory(precision=tau)
    q0 = 0.2 * rs.standard_normal((Cs, K))
    p0, u = rs.standard_normal((Cs, K)), rs.uniform(size=Cs)
    dt = 0.02 / max(1.0, N / 20.0) / K
    res = []
    for fused in (True, False):
        s = HMCSampler(cond, t(q0), dt, int(rs.randint(1, 12)) if False else 5,
                       variable_name='coefficients')
        s.fused_transition = fused
        o = s.sample(p0=t(p0), u=t(u))
        res.append((o.cpu().numpy(), s.last_move_accepted.cpu().numpy(),
                    s.last_e_before.cpu().numpy(), s.last_e_after.cpu().numpy()))
    (qf, af, ebf, eaf), (qg, ag, ebg, eag) = res
    if not (np.array_equal(ebf, ebg) and np.array_equal(af, ag)
            and np.allclose(qf, qg, rtol=1e-10, atol=1e-10 * np.abs(qg).max())
            and np.allclose(eaf, eag, rtol=1e-9, atol=0)):
        report('fused polynomial', K=K, N=N, C=Cs, lik=lik_name, prior='coefficients_prior' in priors)
    # ---- Gibbs-within-HMC vs the single-chain numpy restatement ---------------
    if case % 3 == 0:
        K = int(rs.randint(1, 17))
        N = int(rs.randint(2, 129))
        Cg, S, L = int(rs.randint(1, 6)), int(rs.randint(1, 4)), int(rs.randint(1, 9))
        xs = np.linspace(-1.5, 1.5, N)
        ysp = R.polyval(xs, rs.standard_normal(K)) + 0.5 * rs.standard_normal(N)
        post = Posterior({'points': Likelihood('points', ForwardModel(xs, POLYVAL), GaussianErrorModel(ysp))},
                         {'precision_prior': GammaPrior(1.0, 0.2),
                          'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K) * 5)})
        c0 = 0.2 * rs.standard_normal((Cg, K))
        tau0 = rs.uniform(0.5, 2.0, size=Cg)
        p0, u = rs.standard_normal((S, Cg, K)), rs.uniform(size=(S, Cg))
        gdraw = rs.gamma(R.gamma_shape(N, 1.0), size=(S, Cg))
        dt = 0.02 / max(1.0, N / 20.0) / K
        sweep = {'s': 0}
        gips = make_hmc_sampler(post, dt, L, BinfState(dict(coefficients=t(c0), precision=t(tau0))),
                                gamma=lambda sh, n_, d: t(gdraw[sweep['s']]))
        hmc = gips.subsamplers['coefficients']
        got = []
        for si in range(S):
            sweep['s'] = si
            hmc.rng = type('Inject', (), {'normal': staticmethod(lambda shp, d, si=si: t(p0[si])),
                                          'uniform': staticmethod(lambda n_, d, si=si: t(u[si]))})()
            st = gips.sample()
            got.append((st.variables['coefficients'].cpu().numpy().copy(),
                        st.variables['precision'].cpu().numpy().copy(),
                        hmc.last_move_accepted.cpu().numpy().copy()))
        for c in range(Cg):
            ref = RE.gibbs_hmc_chain(xs, ysp, c0[c], tau0[c], dt, L, p0[:, c], u[:, c], gdraw[:, c])
            for si in range(S):
                if not (bool(got[si][2][c]) == bool(ref['accepted'][si])
                        and np.allclose(got[si][0][c], ref['coefficients'][si], rtol=1e-9, atol=1e-10)
                        and abs(got[si][1][c] - ref['precision'][si]) <= 1e-9 * ref['precision'][si]):
                    report('gibbs vs restatement', K=K, N=N, C=Cg, sweep=si, chain=c)
    if case % 20 == 19:
        print('%d cases, %d mismatches, %.0f s' % (case + 1, bad, time.time() - t0), flush=True)
print('done: %d cases, %d mismatches' % (n_cases, bad))
sys.exit(1 if bad else 0)

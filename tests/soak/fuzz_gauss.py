"""Randomised differential test of the fused Gaussian HMC kernels (all layouts:
several chains per wave, one wave per chain, several waves per chain, split
chains, irregular pairwise trees) against the C oracle, bit for bit.
Development aid / soak test:  python tests/soak/fuzz_gauss.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from binf_amd import _native
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
from oracle import c_oracle

dev = torch.device('cuda:0')
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time()
bad = 0
for case in range(n_cases):
    kind = rs.randint(7)
    if kind == 0:
        D = int(rs.randint(1, 130))
    elif kind == 1:
        D = int(rs.randint(130, 1100))
    elif kind == 2:
        D = int(rs.choice([768, 1024]))
    elif kind == 3:
        D = int(rs.randint(1100, 8193))
    elif kind == 4:
        D = int(8 * rs.randint(1, 1025))
    elif kind == 5:
        D = int(rs.randint(7600, 40000))             # ragged 7-level trees, multi-chunk rows (generic tier)
    else:
        D = int(rs.choice([64, 128, 256, 512, 768, 1024]))   # regular trees, one wave per chain
    C = int(rs.choice([1, 2, 3, 7, 8, 9, 31, 64, 65, 200, 1030]))
    if kind == 6:                                    # enough chains to stay off the split kernel:
        C = int(rs.choice([2049, 2100, 2817]))       # the scalar-step-size instantiation when limit == 0
    if D * C > 3e6:
        C = max(1, int(3e6 // D))
    L = int(rs.randint(1, 9))
    n = int(rs.randint(1, 5))
    thin = int(rs.randint(1, n + 1))
    limit = int(rs.choice([0, 0, 2, 10]))
    mode = 'fma' if rs.rand() < 0.3 else 'exact'       # FMA mode: bit for bit the fused oracle
    fma = mode == 'fma'
    k, x0 = (1.0, 0.0) if rs.rand() < 0.5 else (float(rs.uniform(0.2, 3)), float(rs.normal()))
    dt = float(rs.uniform(0.01, 0.6)) / np.sqrt(k)
    q0 = rs.standard_normal((C, D)) / np.sqrt(k) + x0
    p0 = rs.standard_normal((n, C, D))
    u = rs.uniform(size=(n, C))
    s = HMCSampler(IsotropicGaussian(k, x0), torch.from_numpy(q0).to(dev), dt, L,
                   timestep_adaption_limit=limit, variable_name='x', record_energies=True, mode=mode)
    rec = s.sample_n(n, thin=thin, p0=torch.from_numpy(p0).to(dev), u=torch.from_numpy(u).to(dev))
    torch.cuda.synchronize()
    acc = s.accepted_history.cpu().numpy()
    eb, ea = s.last_e_before.cpu().numpy(), s.last_e_after.cpu().numpy()
    q, dtc = q0, np.full(C, dt)
    ok = True
    for i in range(n):
        adapt = (i + 1) < limit
        w = c_oracle.hmc_sample_gauss(q, p0[i], u[i], dtc, L, k, x0, adapt=adapt, nthreads=8, fma=fma)
        ok &= np.array_equal(acc[i], w['accepted'].astype(bool))
        ok &= np.array_equal(eb[i], w['e_before']) and np.array_equal(ea[i], w['e_after'])
        if (i + 1) % thin == 0:
            ok &= np.array_equal(rec[(i + 1) // thin - 1].cpu().numpy(), w['q_out'])
        q, dtc = w['q_out'], w['timestep_out']
    ok &= np.array_equal(s.state.cpu().numpy(), q)
    if limit:
        ok &= np.array_equal(np.broadcast_to(s.timestep.cpu().numpy() if isinstance(s.timestep, torch.Tensor) else s.timestep, (C,)), dtc)
    # the per-step tier (flat 16-byte kernels; odd D and per-chain dt take the other
    # variants) on the same case, first transition
    if rs.rand() < 0.3:
        pdf = IsotropicGaussian(k, x0)
        pdf.native_hmc_spec = lambda name: None
        g = HMCSampler(pdf, torch.from_numpy(q0).to(dev), dt, L, timestep_adaption_limit=limit,
                       variable_name='x', mode=mode)
        xg = g.sample(p0=torch.from_numpy(p0[0]).to(dev), u=torch.from_numpy(u[0]).to(dev))
        w = c_oracle.hmc_sample_gauss(q0, p0[0], u[0], np.full(C, dt), L, k, x0, adapt=1 < limit,
                                      nthreads=8, fma=fma)
        ok &= np.array_equal(xg.cpu().numpy(), w['q_out'])
        ok &= np.array_equal(g.last_e_after.cpu().numpy(), w['e_after'])
        ok &= np.array_equal(g.last_move_accepted.cpu().numpy(), w['accepted'].astype(bool))
    # draws generated in the kernel == sampling from the dump of the same stream
    if _native.gauss_persist_covers(D) and rs.rand() < 0.4:
        seed, mode = int(rs.randint(1 << 30)), ('exact' if rs.rand() < 0.7 else 'fma')
        kw = dict(timestep_adaption_limit=limit, variable_name='x', mode=mode, record_energies=True)
        a = HMCSampler(IsotropicGaussian(k, x0), torch.from_numpy(q0).to(dev), dt, L,
                       rng=DeviceRNG(seed, dev, fused='always'), **kw)
        ra = a.sample_n(n, thin=thin)
        pd, ud = _native.hmc_gauss_rng_draws(n, C, D, seed, 0, dev)
        b = HMCSampler(IsotropicGaussian(k, x0), torch.from_numpy(q0).to(dev), dt, L, **kw)
        rb = b.sample_n(n, thin=thin, p0=pd, u=ud)
        ok &= (ra is None and rb is None) or torch.equal(ra, rb)
        ok &= torch.equal(a.state, b.state) and torch.equal(a.accepted_history, b.accepted_history)
        ok &= torch.equal(a.last_e_after, b.last_e_after)
        ok &= bool(torch.isfinite(pd).all()) and float(pd.abs().max()) < 9.0
    if not ok:
        bad += 1
        print('MISMATCH', dict(D=D, C=C, L=L, n=n, thin=thin, limit=limit, k=k, x0=x0, dt=dt, fma=fma), flush=True)
    if case % 50 == 49:
        print('%d cases, %d mismatches, %.0f s' % (case + 1, bad, time.time() - t0), flush=True)
print('done: %d cases, %d mismatches' % (n_cases, bad))
sys.exit(1 if bad else 0)

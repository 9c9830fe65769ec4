"""Randomised differential test of the pair-distance kernels beyond 256 beads (development aid /
soak): the ring kernels (257..1024 beads, many chains), a wave per 64 x 64 tile (few chains, and
1025..8192 beads at any count), chi^2 by 8192-pair chunks -- at ARBITRARY bead counts (ragged last
blocks, odd and even block counts) and chain counts either side of the switches.
* force with packed targets vs the one-sided loops (1e-13 of the largest component) and, for the
  first chain, vs the numpy restatement oracle/ref_distance.py (1e-10 of the sum's scale);
* tiles (with the workspace) vs ring (workspace withheld through the C ABI), bit for bit, n <= 1024;
* fused leapfrog vs kick / drift around the same gradient, bit for bit, both arithmetic modes;
* log-prob vs numpy (one launch or by chunks -- the library chooses): bit for bit at precision 1
  and 2 (log exact); to an ulp of the N/2 log(tau) term otherwise (device log).
  python tests/soak/fuzz_pairdist_big.py [n_cases] [seed]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from binf_amd import _native
from binf_amd.example.distance import make_distance_likelihood
from oracle import ref_distance as RD

dev = torch.device('cuda:0')
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
L = _native.lib()
pp = lambda x: ctypes.c_void_p(x.data_ptr())
bad = 0
t0 = time.time()


def report(what, **kw):
    global bad
    bad += 1
    print('MISMATCH', what, kw, flush=True)


for case in range(n_cases):
    kind = int(rs.randint(4))
    if kind == 0:
        n = int(rs.randint(257, 1025))                        # ring or tiles by the chain count
    elif kind == 1:
        n = int(64 * rs.randint(5, 17) + rs.choice([-1, 0, 1]))   # either side of a block boundary
        n = max(257, min(1024, n))
    elif kind == 2:
        n = int(rs.randint(1025, 2400))                       # tiles at any chain count
    else:
        n = int(rs.choice([1025, 1088, 1500, 2047, 2048, 2049, 3000]))
    few = bool(rs.randint(2)) or n > 1024
    C = int(rs.randint(1, 13)) if few else int(rs.randint(140, 420))
    if n > 1500:
        C = min(C, 4)
    truth = rs.standard_normal((n, 3)) * 2.0
    ys = np.abs(RD.forward(truth.reshape(-1), n) + 0.05 * rs.standard_normal(n * (n - 1) // 2))
    x = truth.reshape(-1)[None, :] + 0.3 * rs.standard_normal((C, 3 * n))
    lik = make_distance_likelihood(ys, n)
    em = lik.error_model
    ymat, packed = em.ymat_device(dev), em.ypacked_device(dev)
    if packed is None:
        report('no packed targets', n=n)
        continue
    xx = t(x)
    tau = t(rs.uniform(0.5, 3.0, size=C))
    g = _native.pairdist_gauss_grad(xx, ymat, tau, packed=packed)
    one = _native.pairdist_gauss_grad(xx, ymat, tau)
    if float((g - one).abs().max()) > 1e-13 * float(one.abs().max()):
        report('packed vs one-sided force', n=n, C=C)
    w = RD.gradient(x[0], ys, float(tau[0]), n)
    xc = x[0].reshape(n, 3)
    scale = float(tau[0]) * np.abs(xc[:, None, :] - xc[None, :, :]).sum(axis=1).max()
    if np.abs(g[0].cpu().numpy() - w).max() > 1e-10 * max(np.abs(w).max(), scale):
        report('force vs numpy', n=n, C=C)
    # the same chains as part of a batch on the other side of the ring / tiles switch (n <= 1024)
    if n <= 1024:
        need = L.binf_pairdist_tiles_workspace_bytes(C, n)
        if need > 0:
            # the wrapper took the tiles: withhold the workspace -> ring kernels, same bits
            out = torch.empty_like(xx)
            rc = L.binf_pairdist_gauss_grad_packed_f64(pp(xx), pp(ymat), pp(packed), 0.0, pp(tau), pp(out), C, n,
                                                       None, 0, _native.stream_handle(dev))
            if rc != 0 or not torch.equal(out, g):
                report('tiles vs ring', n=n, C=C, rc=rc)
        else:
            # the wrapper took the ring kernels: the first three chains alone take the tiles
            k = min(3, C)
            gk = _native.pairdist_gauss_grad(xx[:k].contiguous(), ymat, tau[:k].contiguous(), packed=packed)
            if not torch.equal(gk, g[:k]):
                report('ring vs tiles (small batch)', n=n, C=C)
    # fused leapfrog == the per-step sequence around the same gradient
    p0 = t(rs.standard_normal((C, 3 * n)))
    Ls, dt = int(rs.randint(1, 4)), 2e-3
    prior = (0.05, 0.1) if rs.randint(2) else None
    for mode in (_native.MODE_EXACT, _native.MODE_FMA):
        qa, pa = xx.clone(), p0.clone()
        ok = _native.pairdist_leapfrog(qa, pa, ymat, tau, prior, prior is not None, dt, None, Ls, mode, packed=packed)
        if ok is False:
            report('fused leapfrog refused', n=n, C=C)
            continue
        qb, pb = xx.clone(), p0.clone()

        def force(q):
            f = _native.pairdist_gauss_grad(q, ymat, tau, packed=packed)
            if prior is None:
                return f
            return _native.sum_terms([_native.gauss_grad(q, prior[0], prior[1]), f])
        _native.leapfrog_kick(pb, force(qb), dt, None, half=True, mode=mode)
        _native.leapfrog_drift(qb, pb, dt, None, mode=mode)
        for _ in range(Ls - 1):
            _native.leapfrog_kick_drift(qb, pb, force(qb), dt, None, mode=mode)
        _native.leapfrog_kick(pb, force(qb), dt, None, half=True, mode=mode)
        if not (torch.equal(qa, qb) and torch.equal(pa, pb)):
            report('fused leapfrog', n=n, C=C, L=Ls, mode=mode, prior=prior)
    # log-prob: numpy's bits where log(tau) is exact (tau = 1, 2: the chi^2 and its scaling are
    # numpy's, np.sum order); for any other precision log(tau) is the device library's (and numpy's
    # own log differs between CPUs): an ulp of the N/2 log(tau) term
    t0_ = float(rs.choice([1.0, 2.0]))
    lp = lik.log_prob(coordinates=xx, precision=t0_).cpu().numpy()
    for c in {0, C - 1}:
        if lp[c] != RD.log_prob(x[c], ys, t0_, n):
            report('log-prob vs numpy', n=n, C=C, c=c)
    lpc = lik.log_prob(coordinates=xx, precision=tau).cpu().numpy()
    for c in {0, C - 1}:
        want = RD.log_prob(x[c], ys, float(tau[c]), n)
        logz = abs(len(ys) * 0.5 * np.log(float(tau[c])))
        if abs(lpc[c] - want) > 4e-16 * max(logz, abs(want)):
            report('log-prob, precision per chain', n=n, C=C, c=c, got=float(lpc[c]), want=float(want))
    if (case + 1) % 10 == 0:
        print('%d cases, %d mismatches, %.0f s' % (case + 1, bad, time.time() - t0), flush=True)
print('done: %d cases, %d mismatches, %.0f s' % (n_cases, bad, time.time() - t0), flush=True)
sys.exit(1 if bad else 0)

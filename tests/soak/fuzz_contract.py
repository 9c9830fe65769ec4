"""Randomised differential test of the generic chain-rule contraction (csrc/jacobian.hip,
binf/pdf/likelihoods.py:148-155) either side of its switches (development aid / soak): a shared
Jacobian with 16-chain workgroups (small batches) and 32-chain, 8-wave workgroups (from 8192 chains x
1024 data points up); row tiles of 16 / 32 / 48 / 64 rows and more than 64 rows; ragged last tiles.
* rows of a big batch recomputed as small batches: bit for bit (a chain's sums depend on (K, N) only);
* against numpy within 1e-10 * sum |J||r| (the reference's BLAS order is not reproducible);
* a Jacobian per chain: the same two checks.
  python tests/soak/fuzz_contract.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from binf_amd import _native

dev = torch.device('cuda:0')
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t0 = time.time()


def report(what, **kw):
    global bad
    bad += 1
    print('MISMATCH', what, kw, flush=True)


for case in range(n_cases):
    big = bool(rs.randint(2))
    K = int(rs.choice([1, 4, 16, 17, 32, 33, 48, 49, 64, 65, 100])) if rs.randint(2) else int(rs.randint(1, 70))
    if big:
        C, N = int(rs.randint(8192, 9300)), int(rs.randint(1024, 2600))
    else:
        C, N = int(rs.randint(1, 9000)), int(rs.randint(1, 1400))
    g = torch.Generator(device='cpu').manual_seed(int(rs.randint(1 << 30)))
    J = torch.randn(K, N, dtype=torch.float64, generator=g).to(dev)
    r = torch.randn(C, N, dtype=torch.float64, generator=g).to(dev)
    full = _native.jacobian_contract(J, r)
    for _ in range(3):
        lo = int(rs.randint(0, C))
        hi = min(C, lo + int(rs.randint(1, 60)))
        part = _native.jacobian_contract(J, r[lo:hi].contiguous())
        if not torch.equal(part, full[lo:hi]):
            report('shared J: batch independence', K=K, N=N, C=C, lo=lo, hi=hi)
    rows = np.unique(np.concatenate([[0, C - 1], rs.randint(0, C, size=6)]))
    Jn, rn = J.cpu().numpy(), r[rows].cpu().numpy()
    want, bound = rn.dot(Jn.T), 1e-10 * np.abs(rn).dot(np.abs(Jn).T)
    if not np.all(np.abs(full[rows].cpu().numpy() - want) <= bound + 1e-300):
        report('shared J vs numpy', K=K, N=N, C=C)
    # a Jacobian per chain (bounded size)
    Cb = int(min(C, max(1, (1 << 25) // max(1, K * N))))
    Jb = torch.randn(Cb, K, N, dtype=torch.float64, generator=g).to(dev)
    fb = _native.jacobian_contract(Jb, r[:Cb].contiguous())
    c = int(rs.randint(0, Cb))
    one = _native.jacobian_contract(Jb[c:c + 1].contiguous(), r[c:c + 1].contiguous())
    if not torch.equal(one, fb[c:c + 1]):
        report('J per chain: batch independence', K=K, N=N, C=Cb, c=c)
    wantb = Jb[c].cpu().numpy().dot(r[c].cpu().numpy())
    boundb = 1e-10 * np.abs(Jb[c].cpu().numpy()).dot(np.abs(r[c].cpu().numpy()))
    if not np.all(np.abs(fb[c].cpu().numpy() - wantb) <= boundb + 1e-300):
        report('J per chain vs numpy', K=K, N=N, C=Cb, c=c)
    if (case + 1) % 10 == 0:
        print('%d cases, %d mismatches, %.0f s' % (case + 1, bad, time.time() - t0), flush=True)
print('done: %d cases, %d mismatches, %.0f s' % (n_cases, bad, time.time() - t0), flush=True)
sys.exit(1 if bad else 0)

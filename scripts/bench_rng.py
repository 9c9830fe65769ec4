"""Device time of the stand-alone draw generators (csrc/rng.hip -- np.random.normal / uniform /
gamma of hmc.py:146,151 and example/samplers.py:47 in throughput mode) and of the RWMC subsampler's
two kernels (csrc/rwmc.hip -- example/samplers.py:81-90), with the roofline each is bound by:
a generator writes 8 B per draw and reads nothing (HBM write stream; the counter-based Philox rounds
and the transform are its VALU work), the RWMC kernels are per-chain scalar work.
    python scripts/bench_rng.py            -> one JSON line"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native

dev = torch.device('cuda:0')
HBM_PEAK_GBS = 8000.0


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


res = {}
for n in (4096 * 1024, 1 << 26):
    out = torch.empty(n, dtype=torch.float64, device=dev)
    u = torch.empty(4096, dtype=torch.float64, device=dev)
    reps = 200 if n < (1 << 24) else 40
    row = {}
    for kind, shape in (('uniform', None), ('normal', None), ('normal_zig', None), ('gamma', 2.5), ('gamma', 11.0)):
        s = timed(lambda i=0: _native.rng_fill(kind, out, 7, 1 + i, shape=shape), reps)
        row[kind + ('' if shape is None else '_shape_%g' % shape)] = {
            'us': s * 1e6, 'G_draws_per_s': n / s / 1e9, 'GB_per_s_written': 8.0 * n / s / 1e9,
            'hbm_write_frac': 8.0 * n / s / 1e9 / HBM_PEAK_GBS}
    s = timed(lambda i=0: _native.rng_fill_normal_zig_uniform(out, u, 7, 1 + i, 900 + i), reps)
    row['normal_zig_uniform (one launch: p0 [n] + u [4096])'] = {
        'us': s * 1e6, 'G_draws_per_s': n / s / 1e9, 'hbm_write_frac': 8.0 * n / s / 1e9 / HBM_PEAK_GBS}
    res['%d draws' % n] = row
    del out

# RWMC: one precision per chain (K = 1), proposal + Metropolis test, draws on the device
for C in (4096, 1 << 20):
    state = torch.rand(C, 1, dtype=torch.float64, device=dev) + 1.0
    prop = torch.empty_like(state)
    nxt = torch.empty_like(state)
    lp_old = torch.randn(C, dtype=torch.float64, device=dev)
    lp_new = torch.randn(C, dtype=torch.float64, device=dev)
    acc = torch.empty(C, dtype=torch.uint8, device=dev)
    nacc = torch.zeros(C, dtype=torch.int64, device=dev)
    sp = timed(lambda i=0: _native.rwmc_propose(state, 0.3, seed=3, offset=2 * i, out=prop), 300)
    sa = timed(lambda i=0: _native.rwmc_accept(prop, state, lp_old, lp_new, nxt, acc, nacc, seed=3,
                                               offset=2 * i + 1), 300)
    # bytes: propose reads 8 and writes 8 per chain; accept reads 8 * 4, writes 8 + 1 + 8 (+ 8 read)
    res['rwmc, %d chains x 1' % C] = {
        'propose_us': sp * 1e6, 'accept_us': sa * 1e6,
        'propose_GB_per_s': 16.0 * C / sp / 1e9, 'accept_GB_per_s': 57.0 * C / sa / 1e9,
        'note': 'launch-bound at 4096 chains (a dispatch is ~5 us on this chip)'}
print(json.dumps(res))

"""Device time of the generic chain-rule contraction (csrc/jacobian.hip, binf_jacobian_contract_f64 --
Likelihood._evaluate_gradient's dfm.dot(emgrad), binf/pdf/likelihoods.py:148-155) for forward models
without a fused kernel: a Jacobian shared by all chains (linear models: f64 MFMA tiles, flops) and a
Jacobian per chain (streamed once: HBM bytes), and of binf_sum_terms_f64 (Posterior's sum of its
components, posteriors.py:147-151).     python scripts/bench_jacobian.py   -> one JSON line"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native

dev = torch.device('cuda:0')
HBM_PEAK_GBS, MFMA_F64_PEAK_TF = 8000.0, 78.6


def timed(fn, reps=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


res = {}
for C, K, N in ((8192, 33, 16384), (4096, 33, 16384), (4096, 8, 1000), (65536, 4, 20)):
    J = torch.randn(K, N, dtype=torch.float64, device=dev)
    r = torch.randn(C, N, dtype=torch.float64, device=dev)
    s = timed(lambda: _native.jacobian_contract(J, r))
    flops, byts = 2.0 * C * K * N, 8.0 * (C * N + K * N + C * K)
    res['shared J: %d chains, K=%d, N=%d' % (C, K, N)] = {
        'us': s * 1e6, 'TFLOPs': flops / s / 1e12, 'mfma_frac': flops / s / 1e12 / MFMA_F64_PEAK_TF,
        'GB_per_s (emgrad read once)': byts / s / 1e9, 'hbm_frac': byts / s / 1e9 / HBM_PEAK_GBS}
for C, K, N in ((4096, 33, 1024), (512, 33, 16384), (16384, 8, 256), (65536, 4, 20)):
    J = torch.randn(C, K, N, dtype=torch.float64, device=dev)
    r = torch.randn(C, N, dtype=torch.float64, device=dev)
    s = timed(lambda: _native.jacobian_contract(J, r))
    byts = 8.0 * (C * K * N + C * N + C * K)
    res['J per chain: %d chains, K=%d, N=%d' % (C, K, N)] = {
        'us': s * 1e6, 'GB_per_s': byts / s / 1e9, 'hbm_frac': byts / s / 1e9 / HBM_PEAK_GBS,
        'bytes': byts}
for C, D, T in ((4096, 1024, 3), (8192, 33, 3), (256, 768, 2)):
    terms = [torch.randn(C, D, dtype=torch.float64, device=dev) for _ in range(T)]
    s = timed(lambda: _native.sum_terms(terms), 200)
    byts = 8.0 * C * D * (T + 1)
    res['sum of %d terms [%d x %d]' % (T, C, D)] = {'us': s * 1e6, 'GB_per_s': byts / s / 1e9,
                                                    'hbm_frac': byts / s / 1e9 / HBM_PEAK_GBS}
print(json.dumps(res))

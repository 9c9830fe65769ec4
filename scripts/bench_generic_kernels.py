"""The generic per-step tier's kernels one by one at C2's shape (4096 x 1024) and
two others: time per launch (HIP events) and the HBM rate of the bytes each one
has to move; plus the chunked long-chain kernel against the per-step tier.
Development aid / evidence for DESIGN.md 4.2; run under rocprofv3 --kernel-trace
--stats for the per-kernel rows in profiles/."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
dev = torch.device('cuda:0')


def timed(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


out = {}
for C, D in ((4096, 1024), (8192, 33), (64, 20000)):
    q = torch.randn((C, D), dtype=torch.float64, device=dev)
    p = torch.randn((C, D), dtype=torch.float64, device=dev)
    g = torch.randn((C, D), dtype=torch.float64, device=dev)
    q2 = torch.empty_like(q)
    eb = torch.randn(C, dtype=torch.float64, device=dev)
    ea = eb + 0.1 * torch.randn(C, dtype=torch.float64, device=dev)
    u = torch.rand(C, dtype=torch.float64, device=dev)
    acc = torch.empty(C, dtype=torch.uint8, device=dev)
    nacc = torch.zeros(C, dtype=torch.int64, device=dev)
    dtc = torch.full((C,), 1e-3, dtype=torch.float64, device=dev)
    n = C * D
    r = {}
    for name, fn, nbytes in (
            ('kick_drift', lambda: _native.leapfrog_kick_drift(q, p, g, 1e-3), 40.0 * n),
            ('kick_drift_per_chain_dt', lambda: _native.leapfrog_kick_drift(q, p, g, 0.0, dtc), 40.0 * n),
            ('kick', lambda: _native.leapfrog_kick(p, g, 1e-3), 24.0 * n),
            ('drift', lambda: _native.leapfrog_drift(q, p, 1e-3), 24.0 * n),
            ('gauss_grad', lambda: _native.gauss_grad(q, 2.5, 0.3, out=g), 16.0 * n),
            ('row_sumsq', lambda: _native.row_sum(p, _native.ROW_SUMSQ, scale=0.5), 8.0 * n),
            ('accept_select', lambda: _native.accept_select(q, p, eb, ea, u, q2, acc, nacc, None,
                                                            False, 1.05, 0.95), 16.0 * n)):
        t = timed(fn)
        r[name] = {'us': t * 1e6, 'GBps': nbytes / t / 1e9}
    out['%dx%d' % (C, D)] = r

# long chains: the chunked fused kernel vs the per-step tier on the same PDF
for C, D, L in ((64, 20000, 20), (4096, 16384, 20)):
    q0 = torch.randn((C, D), dtype=torch.float64, device=dev)
    p0 = torch.randn((C, D), dtype=torch.float64, device=dev)
    u = torch.rand(C, dtype=torch.float64, device=dev)
    fused = HMCSampler(IsotropicGaussian(), q0, 0.01, L, variable_name='x')
    pdf = IsotropicGaussian()
    pdf.native_hmc_spec = lambda name: None           # force the per-step tier
    step = HMCSampler(pdf, q0, 0.01, L, variable_name='x')
    tf = timed(lambda: fused.sample(p0=p0, u=u), 10, 2)
    ts = timed(lambda: step.sample(p0=p0, u=u), 5, 1)
    out['long_%dx%d_L%d' % (C, D, L)] = {
        'chunked_fused_ms': tf * 1e3, 'per_step_tier_ms': ts * 1e3,
        'chunked_chain_steps_per_s': C * L / tf,
        'chunked_algorithmic_GBps': (24.0 * D + 25.0) * C / tf / 1e9}
print(json.dumps(out))

"""Fused pair-distance leapfrog: launch time against the trajectory length, which
separates the once-per-launch part (target distances to registers) from the cost of
one force evaluation (development aid; numbers in DESIGN.md section 4.4)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
dev = torch.device('cuda:0')
rs = np.random.RandomState(0)


def timed(fn, reps):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


# settle the clocks
z = torch.randn(4096, 4096, device=dev)
t0 = time.time()
while time.time() - t0 < 0.5:
    z = z * 1.0000001
torch.cuda.synchronize()
for n in (256, 200, 48):
    truth = rs.standard_normal((n, 3)) * 2.0
    d = np.sqrt(((truth[:, None, :] - truth[None, :, :]) ** 2).sum(-1))
    ymat = torch.from_numpy(np.abs(d + 0.05 * rs.standard_normal((n, n)))).to(dev)
    for C in (16, 256, 512, 1024, 2048):
        x0 = torch.from_numpy(truth.reshape(-1)[None, :] + 0.1 * rs.standard_normal((C, 3 * n))).to(dev)
        p0 = torch.from_numpy(rs.standard_normal((C, 3 * n))).to(dev)
        for packed in (None, _native.pairdist_pack_targets(ymat)):
            res = {}
            for L in (1, 20):
                q, p = x0.clone(), p0.clone()
                res[L] = timed(lambda: _native.pairdist_leapfrog(q, p, ymat, 4.0, (0.01, 0.0), True,
                                                                 1e-4, None, L, packed=packed), 100)
            per_eval = (res[20] - res[1]) / 19
            tg = timed(lambda: _native.pairdist_gauss_grad(x0, ymat, 4.0, packed=packed), 100)
            print('n=%3d C=%5d %s leapfrog L=1 %.1f us, L=20 %.1f us -> %.2f us per force evaluation, '
                  '%.1f us once per launch; force-only kernel %.1f us; %.2e pairs/s in the trajectory'
                  % (n, C, 'matrix' if packed is None else 'packed', res[1] * 1e6, res[20] * 1e6,
                     per_eval * 1e6, (res[1] - 2 * per_eval) * 1e6, tg * 1e6, C * n * (n - 1) / per_eval))

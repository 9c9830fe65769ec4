"""Force kernel of the pair-distance model vs chain count (development aid)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
dev = torch.device('cuda:0')
n = 256
rs = np.random.RandomState(0)
truth = rs.standard_normal((n, 3)) * 2.0
d = np.sqrt(((truth[:, None, :] - truth[None, :, :]) ** 2).sum(-1))
ymat = torch.from_numpy(np.abs(d + 0.05 * rs.standard_normal((n, n)))).to(dev)
for C in (16, 64, 128, 256, 512, 1023, 1024, 2048):
    x = torch.from_numpy(truth.reshape(-1)[None, :] + 0.1 * rs.standard_normal((C, 3 * n))).to(dev)
    for _ in range(5): _native.pairdist_gauss_grad(x, ymat, 4.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): _native.pairdist_gauss_grad(x, ymat, 4.0)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 50
    print('C=%5d  %.1f us  %.2e pairs/s' % (C, t * 1e6, C * n * n / t))

"""sqrt_rn (pairdist.hip) against numpy's sqrt through the forward kernel: pairs of beads whose
squared distance covers zero, subnormals, the 2^-767 switch, ordinary values and the top of the
range.  Prints the number of differing results (must be 0)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
dev = torch.device('cuda:0')
rs = np.random.RandomState(0)
N = 1 << 20
# bead 0 at the origin, bead 1 at (a, 0, 0): s = a*a exactly representable cases plus random 3-D
ex = rs.uniform(-540, 511, N)
a = np.ldexp(rs.uniform(1, 2, N), ex.astype(np.int64))
a[:8] = [0.0, 5e-324, 1e-200, 2.0 ** -383.5, 2.0 ** -384, 2.0 ** -383, 1e154, 1.3e154]
x = np.zeros((N, 6)); x[:, 3] = a
x[N // 2:, 4] = a[N // 2:] * rs.uniform(0, 1, N - N // 2); x[N // 2:, 5] = a[N // 2:] * rs.uniform(0, 1, N - N // 2)
I = torch.zeros(1, dtype=torch.int32, device=dev); J = torch.ones(1, dtype=torch.int32, device=dev)
bad = 0
for lo in range(0, N, 65535):
    xb = x[lo:lo + 65535]
    got = _native.pairdist_forward(torch.from_numpy(xb).to(dev), I, J).cpu().numpy()[:, 0]
    dd = xb[:, 0:3] - xb[:, 3:6]
    with np.errstate(over='ignore', under='ignore'):
        want = np.sqrt((dd[:, 0] ** 2 + dd[:, 1] ** 2) + dd[:, 2] ** 2)
    bad += int((got.view(np.int64) != want.view(np.int64)).sum())
print('values', N, 'differing', bad)

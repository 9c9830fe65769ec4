import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binf_amd import _native
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
from oracle import c_oracle, ref_numpy as R
dev = torch.device('cuda:0')
for D in (7975, 8164, 7757, 8143, 8192, 7000, 7500, 7700, 7750):
    H = _native.pairwise_tree_height(D)
    leaves, _ = R.pairwise_leaves(D)
    rs = np.random.RandomState(D)
    C, L, dt = 3, 4, 0.2
    q0 = rs.standard_normal((C, D)); p0 = rs.standard_normal((1, C, D)); u = rs.uniform(size=(1, C))
    s = HMCSampler(IsotropicGaussian(1.0, 0.0), torch.from_numpy(q0).to(dev), dt, L, variable_name='x', record_energies=True)
    fused = s._fused_spec('x', D) is not None
    s.sample_n(1, p0=torch.from_numpy(p0).to(dev), u=torch.from_numpy(u).to(dev))
    w = c_oracle.hmc_sample_gauss(q0, p0[0], u[0], dt, L, nthreads=4)
    eb = s.last_e_before.cpu().numpy().reshape(-1); ea = s.last_e_after.cpu().numpy().reshape(-1)
    print(D, 'H', H, 'nleaves', len(leaves), 'lens', sorted(set(l for _, l in leaves)), 'fused', fused,
          'q', np.array_equal(s.state.cpu().numpy(), w['q_out']), 'eb', np.array_equal(eb, w['e_before']), 'ea', np.array_equal(ea, w['e_after']),
          'rowsum', np.array_equal(_native.row_sum(torch.from_numpy(q0).to(dev)).cpu().numpy(), q0.sum(1)))

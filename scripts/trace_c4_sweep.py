"""One C4 Gibbs sweep (4096 chains, K = 33, N = 16384) for rocprofv3 --kernel-trace: which kernels a
sweep is made of and what they cost.  rocprofv3 --kernel-trace --stats -- python3 scripts/trace_c4_sweep.py"""
import os, sys, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'scripts'))
import bench_extra
dev = torch.device('cuda:0')
print(bench_extra.c4_gibbs(dev))

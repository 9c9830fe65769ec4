"""Pair-distance chi^2: the variants against each other, one child process per setting
(BINF_PD_LOGP_ROWS = 1 / 2 chains per workgroup; BINF_PD_LOGP_LDS_TREE=1 = the generic block
reduction with its tree through LDS): time per evaluation and a digest of the result bits
(all settings must agree bit for bit).  Usage: python scripts/probe_pd_rows.py [n_beads]"""
import hashlib, os, subprocess, sys, json

def child(n):
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from binf_amd import _native
    dev = torch.device('cuda:0')
    I, J = np.triu_indices(n, 1)
    ti = torch.from_numpy(I.astype(np.int32)).to(dev); tj = torch.from_numpy(J.astype(np.int32)).to(dev)
    ys = torch.from_numpy(np.random.RandomState(3).uniform(0.5, 3.0, I.size)).to(dev)
    res = {}
    for C in (255, 1024, 2048, 4097, 16384):
        x = torch.from_numpy(np.random.RandomState(C).standard_normal((C, 3 * n))).to(dev)
        out = _native.pairdist_gauss_logp(x, ti, tj, ys, 2.5)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        for _ in range(3): _native.pairdist_gauss_logp(x, ti, tj, ys, 2.5)
        e0.record()
        for _ in range(20): _native.pairdist_gauss_logp(x, ti, tj, ys, 2.5)
        e1.record(); torch.cuda.synchronize()
        res[C] = (round(e0.elapsed_time(e1) / 20 * 1e3, 1), hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12])
    print(json.dumps(res))

if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[2] == 'child':
        child(int(sys.argv[1]))
    else:
        n = sys.argv[1] if len(sys.argv) > 1 else '256'
        for name, env in (('lds tree', {'BINF_PD_LOGP_LDS_TREE': '1'}), ('rows 1', {'BINF_PD_LOGP_ROWS': '1'}),
                          ('rows 2', {'BINF_PD_LOGP_ROWS': '2'}), ('default', {})):
            r = subprocess.run([sys.executable, __file__, n, 'child'], env=dict(os.environ, **env),
                               capture_output=True, text=True)
            print('%-9s' % name, r.stdout.strip() or r.stderr[-400:], flush=True)

# L1 -> L2 read traffic of the pair-distance force kernels: the every-pair-once scheme
# (n = 256) against the one-sided loop (n = 288, the nearest size it still serves), 256
# chains each.  Counters per dispatch, averaged; separate passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/pds.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from binf_amd import _native
dev = torch.device('cuda:0'); C = 256
rs = np.random.RandomState(0)
for n in (256, 288):
    truth = rs.standard_normal((n, 3)) * 2.0
    d = np.sqrt(((truth[:, None, :] - truth[None, :, :]) ** 2).sum(-1))
    ymat = torch.from_numpy(np.abs(d + 0.05 * rs.standard_normal((n, n)))).to(dev)
    x = torch.from_numpy(truth.reshape(-1)[None, :] + 0.1 * rs.standard_normal((C, 3 * n))).to(dev)
    for _ in range(10): _native.pairdist_gauss_grad(x, ymat, 4.0)
torch.cuda.synchronize()
PY
for SET in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_LATENCY_sum SQ_WAVES"; do
rm -rf $R/gpurun_out/pmcpds
rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/gpurun_out/pmcpds -- python3 /tmp/pds.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ['GRAFT_REPO_ROOT']
fs = glob.glob(R + '/gpurun_out/pmcpds/*/*counter_collection.csv')
acc = collections.defaultdict(list)
if fs:
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        if 'pairdist_grad' in k:
            acc[('sym n=256' if 'sym' in k else 'one-sided n=288', r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(k[0], k[1], round(sum(v) / len(v)))
PY
done

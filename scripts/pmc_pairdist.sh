# LDS / VALU / wait counters of the pair-distance force kernel at 256 chains, for the
# in-tree library and (if present) scripts/variants/libbinf_prev.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/pdk.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from binf_amd import _native
dev = torch.device('cuda:0'); n = 256; C = 256
rs = np.random.RandomState(0)
truth = rs.standard_normal((n, 3)) * 2.0
d = np.sqrt(((truth[:, None, :] - truth[None, :, :]) ** 2).sum(-1))
ymat = torch.from_numpy(np.abs(d + 0.05 * rs.standard_normal((n, n)))).to(dev)
x = torch.from_numpy(truth.reshape(-1)[None, :] + 0.1 * rs.standard_normal((C, 3 * n))).to(dev)
for _ in range(20): _native.pairdist_gauss_grad(x, ymat, 4.0)
torch.cuda.synchronize()
PY
for LIB in HEAD prev; do
if [ $LIB = prev ]; then export BINF_LIB_OVERRIDE=$R/scripts/variants/libbinf_prev.so; [ -f $BINF_LIB_OVERRIDE ] || continue; fi
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAIT_ANY"; do
rm -rf $R/gpurun_out/pmcpd
rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/gpurun_out/pmcpd -- python3 /tmp/pdk.py > /dev/null 2>&1
python3 - $LIB <<'PY'
import csv,glob,os,collections,sys
R=os.environ['GRAFT_REPO_ROOT']
fs=glob.glob(R+'/gpurun_out/pmcpd/*/*counter_collection.csv')
acc=collections.defaultdict(list)
if fs:
    for r in csv.DictReader(open(fs[0])):
        if 'pairdist_grad' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[1], {k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
done; done

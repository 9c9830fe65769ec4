cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/polyk.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from binf_amd import _native
from binf_amd.example.likelihood import POLYVAL, ForwardModel
dev = torch.device('cuda:0'); C, K, N = 8192, 33, 16384
xs = np.linspace(-1, 1, N); ys = np.random.RandomState(9).standard_normal(N)
q0 = torch.from_numpy(np.random.RandomState(8).standard_normal((C, K))).to(dev)
fwm = ForwardModel(xs, POLYVAL); A = fwm.design_matrix(K, dev); ty = torch.from_numpy(ys).to(dev)
for _ in range(10): _native.poly_gauss_grad(q0, A, ty, 2.5)
torch.cuda.synchronize()
PY
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmcpoly_a -- python /tmp/polyk.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcpoly_b -- python /tmp/polyk.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ['GRAFT_REPO_ROOT']
for d in ['pmcpoly_a','pmcpoly_b']:
    fs=glob.glob(R+'/gpurun_out/'+d+'/*/*counter_collection.csv')
    if not fs: print(d,'none'); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'poly_grad_mfma' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d,{k: round(sum(v)/len(v)) for k,v in acc.items()})
PY

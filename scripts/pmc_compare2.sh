cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export BINF_GAUSS_NCH=1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc2_a -- $R/scripts/membench 4096 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc2_b -- $R/scripts/membench 4096 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmc2_c -- $R/scripts/membench 4096 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ['GRAFT_REPO_ROOT']
pats=['hmc_gauss','k_real<true, false>','k_real<true, true>']
for d in ['pmc2_a','pmc2_b']:
    fs=glob.glob(R+'/gpurun_out/'+d+'/*/*counter_collection.csv')
    rows=list(csv.DictReader(open(fs[0])))
    for pat in pats:
        acc=collections.defaultdict(list)
        for r in rows:
            if pat in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        # the W=20 dispatches are the later half of each kernel's dispatches
        print(d,pat,{k: round(sum(v[len(v)//2:])/max(1,len(v[len(v)//2:]))) for k,v in acc.items()})
fs=glob.glob(R+'/gpurun_out/pmc2_c/*/*kernel_trace.csv')
rows=list(csv.DictReader(open(fs[0])))
for pat in pats:
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if pat in r['Kernel_Name']]
    h=d[len(d)//2:]
    print('trace',pat,'W=20 mean %.2f us (n=%d); W=1 mean %.2f'%(sum(h)/len(h),len(h),sum(d[:len(d)//2])/(len(d)//2)), 'vgpr',[r['VGPR_Count'] for r in rows if pat in r['Kernel_Name']][0])
PY

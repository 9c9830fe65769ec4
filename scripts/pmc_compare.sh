cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PMC="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
PMC2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
BINF_GAUSS_NCH=1 rocprofv3 --pmc $PMC --output-format csv -d $R/gpurun_out/pmc_lib1 -- $R/scripts/libbench 4096 20 0 > /dev/null 2>&1
BINF_GAUSS_NCH=1 rocprofv3 --pmc $PMC2 --output-format csv -d $R/gpurun_out/pmc_lib2 -- $R/scripts/libbench 4096 20 0 > /dev/null 2>&1
rocprofv3 --pmc $PMC --output-format csv -d $R/gpurun_out/pmc_mb1 -- $R/scripts/membench 4096 > /dev/null 2>&1
rocprofv3 --pmc $PMC2 --output-format csv -d $R/gpurun_out/pmc_mb2 -- $R/scripts/membench 4096 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ['GRAFT_REPO_ROOT']
for d,pat in [('pmc_lib1','hmc_gauss'),('pmc_lib2','hmc_gauss'),('pmc_mb1','k_work<20, true, true>'),('pmc_mb2','k_work<20, true, true>')]:
    fs=glob.glob(R+'/gpurun_out/'+d+'/*/*counter_collection.csv')
    if not fs: print(d,'no file'); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if pat in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d, {k: round(sum(v)/len(v)) for k,v in acc.items()}, 'n=',len(next(iter(acc.values()))) if acc else 0)
PY

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in 20 80; do
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmcp_a$L -- python $R/bench.py --steps 64 --warmup 64 --fuse 64 --nsteps $L --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcp_b$L -- python $R/bench.py --steps 64 --warmup 64 --fuse 64 --nsteps $L --no-cpu-baseline > /dev/null 2>&1
python3 - $L <<'PY'
import csv,glob,os,collections,sys
R=os.environ['GRAFT_REPO_ROOT']; L=sys.argv[1]
for d in ['pmcp_a'+L,'pmcp_b'+L]:
    fs=glob.glob(R+'/gpurun_out/'+d+'/*/*counter_collection.csv')
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'persist' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print('L='+L, d, {k: round(sum(v)/len(v)) for k,v in acc.items()}, 'n=',len(next(iter(acc.values()))))
PY
done

# VALU occupancy and clock of the pair-distance force kernels (development aid).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in 256 2048; do
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmcdist_$C -- python $R/scripts/bench_distance.py $C > /dev/null 2>&1 || exit 1
python3 - $C <<'PY'
import csv,glob,os,collections,sys
R=os.environ['GRAFT_REPO_ROOT']; C=sys.argv[1]
d=R+'/gpurun_out/pmcdist_'+C
for kern in ('pairdist_grad', 'pairdist_leapfrog'):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(d+'/*/*counter_collection.csv')[0])):
        if kern in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    if not acc: continue
    m={k: sum(v)/len(v) for k,v in acc.items()}
    rows=[r for r in csv.DictReader(open(glob.glob(d+'/*/*kernel_trace.csv')[0])) if kern in r['Kernel_Name']]
    dur=sum((int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in rows)/len(rows)*1e-9
    cyc=m['GRBM_GUI_ACTIVE']/8
    print('C=%s %-18s %.1f us  clock %.2f GHz  VALU busy %.1f%%  VALU inst/wave %.0f  LDS inst/wave %.0f  LDS busy %.1f%%  waves %d  wave-life %.2f' % (
        C, kern, dur*1e6, cyc/dur*1e-9, 100*m['SQ_ACTIVE_INST_VALU']*4/(1024*cyc), m['SQ_INSTS_VALU']/m['SQ_WAVES'], m['SQ_INSTS_LDS']/m['SQ_WAVES'],
        100*m['SQ_ACTIVE_INST_LDS']*4/(1024*cyc), m['SQ_WAVES'], m['SQ_WAVE_CYCLES']*4/m['SQ_WAVES']/cyc))
PY
done

# Shader clock and VALU occupancy of the persistent kernel per library variant
# (development aid): GRBM_GUI_ACTIVE / duration = clock, SQ_ACTIVE_INST_VALU x 4 /
# (1024 SIMDs x cycles) = VALU pipe busy.  Usage: bash scripts/pmc_clock.sh [libs...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for lib in default "$@"; do
  i=$((i+1))
  if [ "$lib" != default ]; then export BINF_LIB_OVERRIDE=$R/$lib; else unset BINF_LIB_OVERRIDE; fi
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmcclk_$i -- python $R/bench.py --steps 128 --warmup 64 --fuse 64 --no-cpu-baseline > /dev/null 2>&1 || exit 1
  python3 - $i "$lib" <<'PY'
import csv,glob,os,collections,sys
R=os.environ['GRAFT_REPO_ROOT']; i=sys.argv[1]
d=R+'/gpurun_out/pmcclk_'+i
acc=collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(d+'/*/*counter_collection.csv')[0])):
    if 'persist' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
m={k: sum(v)/len(v) for k,v in acc.items()}
rows=[r for r in csv.DictReader(open(glob.glob(d+'/*/*kernel_trace.csv')[0])) if 'persist' in r['Kernel_Name']]
dur=sum((int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in rows)/len(rows)*1e-9
cyc=m['GRBM_GUI_ACTIVE']/8
print('%-62s %.1f us/launch  clock %.2f GHz  VALU busy %.1f%%  VALU inst/wave/transition %.0f  wave-cycles/total %.2f wait_inst %.2f' % (
    sys.argv[2], dur*1e6, cyc/dur*1e-9, 100*m['SQ_ACTIVE_INST_VALU']*4/(1024*cyc), m['SQ_INSTS_VALU']/4096/64,
    m['SQ_WAVE_CYCLES']*4/4096/cyc, m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']))
PY
done

# kernel-trace of bench.py for several variants; prints mean kernel duration
cd /tmp && export TMPDIR=/tmp
for nch in 1 2; do for mode in exact fma; do for L in 1 20; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pv_${nch}_${mode}_${L}
  BINF_GAUSS_NCH=$nch rocprofv3 --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --nsteps $L --mode $mode --no-cpu-baseline > /dev/null 2>&1
  python - "$out" "NCH=$nch $mode L=$L" <<'PY'
import csv,glob,sys,statistics
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'hmc_gauss' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows][20:]
st=[int(r['Start_Timestamp']) for r in rows][20:]
period=(st[-1]-st[0])/1e3/(len(st)-1)
print('%s kernel mean %.2f med %.2f min %.2f us; launch period %.2f us; vgpr %s'%(sys.argv[2],statistics.mean(d),statistics.median(d),min(d),period,rows[0]['VGPR_Count']))
PY
done; done; done

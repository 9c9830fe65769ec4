set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_c
mkdir -p $O
python3 scripts/probe_single_call.py | tee $O/probe.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f1 -- python3 $R/bench.py --fuse 1 --steps 200 --warmup 50 --no-cpu-baseline --no-other-mode --no-extra --no-pmc > $O/f1.json 2>$O/err.txt
python3 - $O <<'PY'
import csv,glob,sys
O=sys.argv[1]
rows=[r for r in csv.DictReader(open(glob.glob(O+'/prof_f1/*/*kernel_trace.csv')[0])) if 'persist' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
g=[(int(rows[i+1]['Start_Timestamp'])-int(rows[i]['End_Timestamp']))/1e3 for i in range(len(rows)-1)]
import statistics as st
print('n=%d kernel mean %.2f us median %.2f; gap mean %.2f median %.2f'%(len(d),st.mean(d[50:]),st.median(d[50:]),st.mean(g[50:]),st.median(g[50:])))
PY

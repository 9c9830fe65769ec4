# Final evidence for a round: full GPU test-suite, smoke, default bench, the
# same command under rocprofv3 --kernel-trace --stats, and the HBM counters in
# separate --pmc passes.  Usage: bash scripts/final_profile.sh <tag>
set -o pipefail
TAG=${1:-r01_d}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $O/pytest_tail.txt || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt || exit 1
python bench.py > $O/bench.json 2> $O/bench.err || exit 1
cat $O/bench.json
python bench.py --fuse 1 --no-cpu-baseline > $O/bench_fuse1.json 2>/dev/null
python bench.py --mode fma --no-cpu-baseline > $O/bench_fma.json 2>/dev/null
python bench.py --thin 64 --no-cpu-baseline > $O/bench_thin64.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --no-cpu-baseline --no-other-mode > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 256 --warmup 64 --no-cpu-baseline --no-other-mode > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 256 --warmup 64 --no-cpu-baseline --no-other-mode > /dev/null 2>&1
python3 - $O <<'PY'
import csv,glob,sys,collections,json,statistics
O=sys.argv[1]
out={}
for d,c in [('pmc_fetch','FETCH_SIZE'),('pmc_write','WRITE_SIZE')]:
    fs=glob.glob(O+'/'+d+'/*/*counter_collection.csv')
    v=[float(r['Counter_Value']) for r in csv.DictReader(open(fs[0])) if 'persist' in r['Kernel_Name'] and r['Counter_Name']==c]
    out[c+'_KB_per_launch']=sum(v)/len(v); out[c+'_n']=len(v)
rows=[r for r in csv.DictReader(open(glob.glob(O+'/prof/*/*kernel_trace.csv')[0])) if 'persist' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
out['rocprof_kernel_mean_us']=statistics.mean(d); out['rocprof_kernel_n']=len(d)
b=json.load(open(O+'/bench.json'))
out['bench_avg_launch_us']=b['roofline']['avg_launch_us']
out['bench_under_rocprof_avg_launch_us']=json.load(open(O+'/bench_under_rocprof.json'))['roofline']['avg_launch_us']
print(json.dumps(out))
json.dump(out,open(O+'/summary.json','w'),indent=1)
c=b['config']; F=b['roofline']['transitions_per_launch']
traffic={'config': {'chains': c['chains_per_gpu'], 'dims': c['n_dims'], 'nsteps': c['leapfrog_steps'],
                    'fuse': F, 'thin': 1, 'mode': 'exact'},
         'FETCH_SIZE_KB_per_launch': out['FETCH_SIZE_KB_per_launch'],
         'WRITE_SIZE_KB_per_launch': out['WRITE_SIZE_KB_per_launch'],
         'correction': 'gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section: reads tallied at half size; confirmed by scripts/pmc_calibrate.sh: 0.5000 / 1.0000 on a 1 GiB elementwise kernel); WRITE_SIZE as is; KB = 1024 B',
         'hbm_bytes_per_transition': (2*out['FETCH_SIZE_KB_per_launch']+out['WRITE_SIZE_KB_per_launch'])*1024/F,
         'collected': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 256 --warmup 64 --no-cpu-baseline; mean over the persist-kernel dispatches (scripts/final_profile.sh)'}
json.dump(traffic,open(O+'/pmc_traffic.json','w'),indent=1)
PY
bash $R/scripts/pmc_calibrate.sh > $O/pmc_calibration.txt 2>&1
cd $R
python scripts/bench_poly.py > $O/bench_poly.json 2>/dev/null
python scripts/bench_distance.py 256 > $O/bench_distance_256.json 2>/dev/null
python scripts/bench_distance.py 2048 > $O/bench_distance_2048.json 2>/dev/null
python scripts/bench_e2e.py > $O/bench_e2e.json 2>/dev/null
tail -n 1 $O/bench_poly.json $O/bench_distance_256.json $O/bench_distance_2048.json $O/bench_e2e.json

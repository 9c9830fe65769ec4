set -o pipefail
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -3 || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
python bench.py > $R/gpurun_out/r1c_bench.json 2> $R/gpurun_out/r1c_bench.err || exit 1
cat $R/gpurun_out/r1c_bench.json
python bench.py --fuse 1 --no-cpu-baseline 2>/dev/null > $R/gpurun_out/r1c_bench_fuse1.json; cat $R/gpurun_out/r1c_bench_fuse1.json | cut -c1-400
python bench.py --mode fma --no-cpu-baseline 2>/dev/null > $R/gpurun_out/r1c_bench_fma.json; cat $R/gpurun_out/r1c_bench_fma.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1c -- python $R/bench.py --steps 128 --no-cpu-baseline > $R/gpurun_out/r1c_bench_prof.json 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_r1c_fetch -- python $R/bench.py --steps 64 --warmup 32 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_r1c_write -- python $R/bench.py --steps 64 --warmup 32 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ['GRAFT_REPO_ROOT']
for d in ['pmc_r1c_fetch','pmc_r1c_write']:
    fs=glob.glob(R+'/gpurun_out/'+d+'/*/*counter_collection.csv')
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'persist' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d,{k:(sum(v)/len(v),len(v)) for k,v in acc.items()})
PY

# Evidence for the multi-sweep Gibbs launch and the device-resident RWMC / Gamma path:
#  1. scripts/bench_gibbs_n.py (loop of sample() vs sample_n) plain and under
#     rocprofv3 --kernel-trace --stats;
#  2. examples/polynomial_fit.py --rwmc (the reference's own wiring, device draws) under
#     rocprofv3 --memory-copy-trace: the number of host-to-device copies must not grow
#     with the number of sweeps (1000 vs 3000 iterations, one sample() per iteration).
# Usage: bash scripts/r03_gibbs_profile.sh <tag>
set -o pipefail
TAG=${1:-r03_h}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
python3 $R/scripts/bench_gibbs_n.py --chains 4096 65536 --out $O/gibbs_n.jsonl || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gibbs -- python3 $R/scripts/bench_gibbs_n.py --chains 4096 > $O/gibbs_n_under_rocprof.jsonl 2>/dev/null || exit 1
for IT in 1000 3000; do
  rocprofv3 --memory-copy-trace --stats --output-format csv -d $O/copy_rwmc_$IT -- python3 $R/examples/polynomial_fit.py --chains 4096 --rwmc --per-launch 1 --iterations $IT --burn-in 500 > $O/example_rwmc_$IT.txt 2>&1 || exit 1
done
rocprofv3 --memory-copy-trace --stats --output-format csv -d $O/copy_rwmc_hostrng -- python3 $R/examples/polynomial_fit.py --chains 4096 --rwmc --host-rng --per-launch 1 --iterations 1000 --burn-in 500 > $O/example_rwmc_hostrng.txt 2>&1 || exit 1
cd $R
python3 - <<PY
import csv, glob, json, os
O = '$O'
out = {}
for tag in ('copy_rwmc_1000', 'copy_rwmc_3000', 'copy_rwmc_hostrng'):
    rows = []
    for f in glob.glob(os.path.join(O, tag, '*', '*memory_copy_trace.csv')):
        rows += list(csv.DictReader(open(f)))
    kinds = {}
    for r in rows:
        k = r.get('Direction') or r.get('Name') or 'copy'
        kinds[k] = kinds.get(k, 0) + 1
    out[tag] = kinds
json.dump(out, open(os.path.join(O, 'memory_copies.json'), 'w'), indent=1)
print(json.dumps(out))
PY

set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_d
mkdir -p $O
python -m pytest tests/test_gpu_hmc_gauss.py tests/test_gpu_edges.py tests/test_gpu_guards.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_tail.txt || exit 1
for M in exact fma; do
python3 bench.py --fuse 1 --mode $M --steps 400 --warmup 100 --no-cpu-baseline --no-other-mode --no-extra --no-pmc > $O/f1_$M.json 2>/dev/null
python3 bench.py --mode $M --no-cpu-baseline --no-other-mode --no-extra --no-pmc > $O/f64_$M.json 2>/dev/null
python3 - $O $M <<'PY'
import json,sys
O,S=sys.argv[1:]
a=json.load(open(O+'/f1_%s.json'%S)); b=json.load(open(O+'/f64_%s.json'%S))
print('%s: fuse1 %.2f us/transition (%.3e)  fuse64 %.2f us/transition (%.3e)'%(S,a['roofline']['avg_transition_us'],a['value'],b['roofline']['avg_transition_us'],b['value']))
PY
done

# stagger sweep for the one-transition-per-launch kernel and the persistent one
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_b
mkdir -p $O
python -m pytest tests/test_gpu_poly.py tests/test_gpu_hmc_gauss.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_tail.txt || exit 1
for S in 0 20 40 70 100 150 220; do
  BINF_GAUSS_STAGGER=$S python3 bench.py --fuse 1 --steps 400 --warmup 100 --no-cpu-baseline --no-other-mode --no-extra --no-pmc > $O/f1_$S.json 2>/dev/null
  BINF_GAUSS_STAGGER=$S python3 bench.py --no-cpu-baseline --no-other-mode --no-extra --no-pmc > $O/f64_$S.json 2>/dev/null
  python3 - $O $S <<'PY'
import json,sys
O,S=sys.argv[1:]
a=json.load(open(O+'/f1_%s.json'%S)); b=json.load(open(O+'/f64_%s.json'%S))
print('stagger %s: fuse1 %.2f us/transition (%.3e)  fuse64 %.2f us/transition (%.3e)'%(S,a['roofline']['avg_transition_us'],a['value'],b['roofline']['avg_transition_us'],b['value']))
PY
done

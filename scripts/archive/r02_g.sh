set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_g
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_hmc_gauss.py tests/test_gpu_guards.py -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_tail.txt || exit 1
python3 scripts/bench_generic_kernels.py > $O/generic_kernels.json; python3 -c "
import json;d=json.load(open('$O/generic_kernels.json'))
for k in d:
    if k.startswith('long'): print(k,d[k])"

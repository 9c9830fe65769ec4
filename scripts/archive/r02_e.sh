set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_e
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_rng.py tests/test_gpu_hmc_gauss.py tests/test_gpu_statistics.py -m gpu -x -q 2>&1 | tail -25 | tee $O/pytest_tail.txt || exit 1
python3 scripts/bench_extra.py > $O/extra.json 2>$O/extra.err; python3 -c "
import json;d=json.load(open('$O/extra.json'));print(json.dumps(d['C2_device_rng']))"
python3 bench.py --no-cpu-baseline --no-extra --no-pmc | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['value'],d['roofline']['avg_transition_us'],d['other_mode']['avg_transition_us'])"

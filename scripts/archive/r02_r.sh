set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 || exit 1
timeout -k 10 200 python3 tests/soak/fuzz_models.py 300 31 2>&1 | tail -2
python3 scripts/bench_extra.py | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print(json.dumps(d['C3']))"

for i in 1 2 3; do python3 - <<'PY'
import sys, torch
sys.path.insert(0, '.')
from scripts import bench_extra
print(bench_extra.c2_device_rng(torch.device('cuda:0')))
PY
done
python3 bench.py --no-cpu-baseline --no-extra --no-pmc --no-other-mode | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('headline',d['value'],d['roofline']['avg_transition_us'])"

# Calibration of the HBM byte counters on a kernel with known traffic: an
# elementwise fp64 scale of a 1 GiB tensor (reads 1 GiB, writes 1 GiB), far
# larger than the 256 MiB Infinity Cache.  Separate --pmc passes, as the guide
# prescribes.  Prints counter / expected.  Usage: bash scripts/pmc_calibrate.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/calib.py <<'PY'
import torch
x = torch.randn(1 << 27, dtype=torch.float64, device='cuda:0')
y = torch.empty_like(x)
for _ in range(6):
    torch.mul(x, 1.5, out=y)
torch.cuda.synchronize()
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_calib_$c -- python /tmp/calib.py > /dev/null 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, os, json
R = os.environ['GRAFT_REPO_ROOT']
out = {'kernel': 'torch.mul(x, 1.5, out=y), 2^27 fp64 elements', 'expected_bytes_each_way': float(1 << 30)}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob(R + '/gpurun_out/pmc_calib_%s/*/*counter_collection.csv' % c)[0]
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(f))
         if r['Counter_Name'] == c and 'elementwise' in r['Kernel_Name']]
    v = v[1:]                                   # skip the first (cold) dispatch
    kb = sum(v) / len(v)
    out[c + '_KB'] = kb
    out[c + '_ratio_to_expected'] = kb * 1024 / float(1 << 30)
print(json.dumps(out))
json.dump(out, open(R + '/gpurun_out/pmc_calibration.json', 'w'), indent=1)
PY

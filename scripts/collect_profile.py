"""Copy the judged summaries of `final_profile.sh <tag>` from gpurun_out/<tag>/
into profiles/<tag>_* (tracked).  Usage: python scripts/collect_profile.py <tag>"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G = os.path.join(ROOT, 'gpurun_out', tag)
P = os.path.join(ROOT, 'profiles')
for name in ('bench', 'bench_under_rocprof', 'bench_fuse1', 'bench_thin64', 'bench_fma',
             'pmc_traffic', 'summary'):
    shutil.copy(os.path.join(G, name + '.json'), os.path.join(P, '%s_%s.json' % (tag, name)))
shutil.copy(os.path.join(ROOT, 'gpurun_out', 'pmc_calibration.json'),
            os.path.join(P, tag + '_pmc_calibration.json'))
for name in ('bench_poly', 'bench_e2e'):
    last = open(os.path.join(G, name + '.json')).read().strip().splitlines()[-1]
    open(os.path.join(P, '%s_%s.json' % (tag, name)), 'w').write(last + '\n')
with open(os.path.join(P, tag + '_bench_distance.json'), 'w') as f:
    for n in ('256', '2048'):
        f.write(open(os.path.join(G, 'bench_distance_%s.json' % n)).read().strip().splitlines()[-1] + '\n')
stats = sorted(glob.glob(os.path.join(G, 'prof', '*', '*kernel_stats.csv')), key=os.path.getmtime)[-1]
rows = list(csv.reader(open(stats)))
for r in rows[1:]:
    if len(r[0]) > 160:
        r[0] = r[0][:157] + '...'
csv.writer(open(os.path.join(P, tag + '_bench_kernel_stats.csv'), 'w')).writerows(rows)
b = json.load(open(os.path.join(G, 'bench.json')))
s = json.load(open(os.path.join(G, 'summary.json')))
print('bench value %.3e frac %.3f; rocprof mean %.1f us (n=%d) vs events %.1f us (that run %.1f us); traffic %.1f MB/transition'
      % (b['value'], b['roofline']['frac'], s['rocprof_kernel_mean_us'], s['rocprof_kernel_n'],
         s['bench_avg_launch_us'], s['bench_under_rocprof_avg_launch_us'],
         json.load(open(os.path.join(G, 'pmc_traffic.json')))['hbm_bytes_per_transition'] / 1e6))

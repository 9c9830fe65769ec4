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
for name in ('bench_poly', 'bench_distance_256', 'bench_distance_2048'):
    last = open(os.path.join(G, name + '.json')).read().strip().splitlines()[-1]
    open(os.path.join(P, '%s_%s.json' % (tag, name)), 'w').write(last + '\n')
for sub, out in (('prof', 'bench_kernel_stats'), ('prof_e2e', 'fused_generator_kernel_stats'),
                 ('prof_poly', 'bench_poly_kernel_stats'), ('prof_dist_256', 'bench_distance_256_kernel_stats'),
                 ('prof_dist_2048', 'bench_distance_2048_kernel_stats')):
    stats = sorted(glob.glob(os.path.join(G, sub, '*', '*kernel_stats.csv')), key=os.path.getmtime)[-1]
    rows = list(csv.reader(open(stats)))
    for r in rows[1:]:
        if len(r[0]) > 160:
            r[0] = r[0][:157] + '...'
    csv.writer(open(os.path.join(P, '%s_%s.csv' % (tag, out)), 'w')).writerows(rows[:25])
for name in ('mfma64_duty.jsonl',):
    if os.path.exists(os.path.join(G, name)):
        shutil.copy(os.path.join(G, name), os.path.join(P, '%s_%s' % (tag, name)))
for name in ('bench_gloo2', 'bench_gloo2_strong'):
    f = os.path.join(G, name + '.json')
    if os.path.exists(f) and open(f).read().strip():
        last = open(f).read().strip().splitlines()[-1]
        open(os.path.join(P, '%s_%s_rehearsal.json' % (tag, name)), 'w').write(last + '\n')
for name, out in (('bench_force_dist_rccl', 'bench_force_dist_rccl_1rank'), ('bench_generic', 'bench_generic_graph')):
    f = os.path.join(G, name + '.json')
    if os.path.exists(f) and open(f).read().strip():
        last = open(f).read().strip().splitlines()[-1]
        open(os.path.join(P, '%s_%s.json' % (tag, out)), 'w').write(last + '\n')
for name in ('pytest_tail.txt', 'smoke.txt'):
    shutil.copy(os.path.join(G, name), os.path.join(P, '%s_%s' % (tag, name)))
b = json.load(open(os.path.join(G, 'bench.json')))
s = json.load(open(os.path.join(G, 'summary.json')))
pk = s['persist_kernel']
print('bench value %.3e frac %.3f hbm_frac_measured %.3f valu_frac %.3f; rocprof mean %.1f us (n=%d) vs events '
      '%.1f us (that run %.1f us); traffic %.1f MB/transition'
      % (b['value'], b['roofline']['frac'], b['roofline']['hbm_frac_measured'], b['roofline']['valu_frac'],
         pk['rocprof_kernel_mean_us'], pk['rocprof_kernel_n'], pk['bench_avg_launch_us'],
         pk['bench_under_rocprof_avg_launch_us'], pk['pmc_bytes_per_transition'] / 1e6))

"""Turn the raw rocprofv3 outputs of final_profile.sh into small JSON summaries
(runs on the GPU box at the end of final_profile.sh; pure csv/json)."""
import collections
import csv
import glob
import json
import statistics
import sys

O = sys.argv[1]


def first(pattern):
    fs = glob.glob(pattern, recursive=True)
    return fs[0] if fs else None


def counters(d, kernel):
    f = first(O + '/' + d + '/**/*counter_collection.csv')
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if kernel in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def durations(d, kernel):
    """kernel durations in dispatch order (us)"""
    f = first(O + '/' + d + '/**/*kernel_trace.csv')
    if not f:
        return []
    rows = [r for r in csv.DictReader(open(f)) if kernel in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    return [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]


out = {}
b = json.load(open(O + '/bench.json'))
F = b['roofline']['transitions_per_launch']
fe, nfe = counters('pmc_fetch', 'persist')
wr, nwr = counters('pmc_write', 'persist')
d = durations('prof', 'persist')
# only the multi-transition launches (a run may also hold one-transition launches of the same kernel)
d = [x for x in d if x > 0.3 * max(d)] if d else d
out['persist_kernel'] = {
    'rocprof_kernel_mean_us': statistics.mean(d) if d else None, 'rocprof_kernel_n': len(d),
    # the last 20 dispatches are bench.py's timed steps (settle and warm-up come before)
    'rocprof_kernel_mean_us_timed_steps': statistics.mean(d[-20:]) if len(d) >= 20 else None,
    'rocprof_first_20_dispatches_us': [round(x, 1) for x in d[:20]],
    'rocprof_kernel_min_us': min(d) if d else None, 'rocprof_kernel_max_us': max(d) if d else None,
    'bench_avg_launch_us': b['roofline']['avg_launch_us'],
    'bench_under_rocprof_avg_launch_us':
        json.load(open(O + '/bench_under_rocprof.json'))['roofline']['avg_launch_us'],
    'bench_live_traffic_bytes_per_launch': b['roofline']['traffic']}
if fe and wr:
    traffic = {'config': {'chains': b['config']['chains_per_gpu'], 'dims': b['config']['n_dims'],
                          'nsteps': b['config']['leapfrog_steps'], 'fuse': F, 'thin': 1,
                          'mode': 'exact'},
               'FETCH_SIZE_KB_per_launch': fe['FETCH_SIZE'], 'WRITE_SIZE_KB_per_launch': wr['WRITE_SIZE'],
               'dispatches': [nfe['FETCH_SIZE'], nwr['WRITE_SIZE']],
               'correction': 'gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md HBM section; calibration '
                             'profiles/r01_h_pmc_calibration.json: 0.5000 / 1.0000); KB = 1024 B',
               'hbm_bytes_per_transition': (2 * fe['FETCH_SIZE'] + wr['WRITE_SIZE']) * 1024 / F,
               'collected': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate '
                            'passes of `bench.py --pmc-child` (scripts/final_profile.sh)'}
    json.dump(traffic, open(O + '/pmc_traffic.json', 'w'), indent=1)
    out['persist_kernel']['pmc_bytes_per_transition'] = traffic['hbm_bytes_per_transition']


def valu(d, kernel, trace_dir=None):
    m, _ = counters(d, kernel)
    dur = durations(trace_dir or d, kernel)
    if not m or not dur:
        return None
    # counters are averaged over all dispatches, so is the duration
    t = statistics.mean(dur) * 1e-6
    cyc = m['GRBM_GUI_ACTIVE'] / 8                     # summed over the 8 XCDs
    r = {'kernel_us': t * 1e6, 'shader_clock_GHz': cyc / t * 1e-9,
         'valu_busy_frac': m['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * cyc),
         'valu_insts_per_wave': m['SQ_INSTS_VALU'] / m['SQ_WAVES'], 'waves': m['SQ_WAVES']}
    if 'SQ_INSTS_LDS' in m:
        r['lds_insts_per_wave'] = m['SQ_INSTS_LDS'] / m['SQ_WAVES']
    return r


out['persist_kernel_valu'] = valu('pmc_valu', 'persist')
out['fused_generator_kernel_valu'] = valu('pmc_e2e', 'persist')
de = durations('prof_e2e', 'persist')
out['fused_generator_kernel'] = {'rocprof_kernel_mean_us': statistics.mean(de) if de else None,
                                 'n': len(de), 'transitions_per_launch': 64,
                                 'us_per_transition': statistics.mean(de) / 64 if de else None,
                                 'us_per_transition_last_40': statistics.mean(de[-40:]) / 64
                                 if len(de) >= 40 else None}
for key, d in (('poly_grad_mfma', 'pmc_poly'), ('poly_grad_mfma_general_kernel', 'pmc_poly_general')):
    m, _ = counters(d, 'poly_grad_mfma')
    dp = durations(d, 'poly_grad_mfma')
    if m and dp:
        t = statistics.mean(dp) * 1e-6
        cyc = m['GRBM_GUI_ACTIVE'] / 8
        out[key] = {'kernel_us_under_pmc': t * 1e6, 'shader_clock_GHz': cyc / t * 1e-9,
                    # busy cycles summed over the 1024 SIMDs / kernel cycles
                    'mfma_busy_frac': m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc),
                    'mfma_insts_per_wave': m['SQ_INSTS_MFMA'] / m['SQ_WAVES'],
                    'valu_insts_per_wave': m['SQ_INSTS_VALU'] / m['SQ_WAVES'],
                    'valu_insts_per_mfma': m['SQ_INSTS_VALU'] / m['SQ_INSTS_MFMA'],
                    'useful_TFLOPs': 4.0 * 33 * 16384 * 8192 / t / 1e12}
# where a C3 sample() spends its time (kernel trace of scripts/bench_poly.py)
f = first(O + '/prof_poly/**/*kernel_stats.csv')
if f:
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    out['poly_bench_kernel_shares'] = [
        {'kernel': r['Name'][:90], 'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) / 1e3,
         'share': float(r['TotalDurationNs']) / tot} for r in rows[:8]]
for C in ('256', '2048'):
    out['pairdist_%s' % C] = {k: valu('pmc_dist_' + C, k) for k in ('pairdist_grad', 'pairdist_leapfrog')}
json.dump(out, open(O + '/summary.json', 'w'), indent=1)
print(json.dumps(out))

"""Condense gpurun_out/<tag>/ (scripts/profile_round.sh) into profiles/<tag>_*.{md,csv,json}."""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
workload = sys.argv[2] if len(sys.argv) > 2 else 'ns'
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)

def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n.split('(')[0]

# 1. kernel stats (all kernels, short names) -> csv
rows = list(csv.DictReader(open(os.path.join(src, 'trace', 'bench_kernel_stats.csv'))))
with open(os.path.join(dst, '%s_kernel_stats.csv' % tag), 'w') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'calls', 'total_ns', 'avg_ns', 'pct', 'min_ns', 'max_ns'])
    for r in rows:
        w.writerow([short(r['Name'])[:120], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
ts = {short(r['Name']): r for r in rows if 'tsamd' in r['Name']}

# 2. PMC passes
pmc = {}
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    p = os.path.join(src, 'pmc_%s' % C, 'pmc_counter_collection.csv')
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        k = short(r['Kernel_Name'])
        pmc.setdefault(k, {}).setdefault(C, []).append(float(r['Counter_Value']))
pmc_rel = {}
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    p = os.path.join(src, 'pmcrel_%s' % C, 'pmc_counter_collection.csv')
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        k = short(r['Kernel_Name'])
        pmc_rel.setdefault(k, {}).setdefault(C, []).append(float(r['Counter_Value']))
bench = None
bj = os.path.join(src, 'bench.json')
if os.path.exists(bj):
    for line in open(bj):
        if line.startswith('{'):
            bench = json.loads(line)

merge = [k for k in pmc if 'spmm_merge_kernel' in k]
traffic = None
lines = ['# %s profile summary (%s workload)' % (tag, workload), '',
         'Source: `scripts/profile_round.sh %s` on one MI355X (gpurun); raw CSVs were under `gpurun_out/%s/`.' % (tag, tag), '',
         '## rocprofv3 --kernel-trace --stats  (`python bench.py`, the default command: headline + control + relabelled leg + C2/C3/C4)', '',
         '| kernel | calls | avg us | total us | % of GPU time |', '|---|---|---|---|---|']
for k, r in ts.items():
    lines.append('| `%s` | %s | %.2f | %.1f | %s |' % (k, r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3, r['Percentage']))
lines += ['', '(the remaining GPU time of that run is ATen input generation: R-MAT sampling, sort/unique, randn; the merge kernel '
          'average above mixes the north-star launches with the control graph, the relabelled leg and config 2)', '']
ph = os.path.join(src, 'trace_headline', 'bench_kernel_stats.csv')
head_avg = None
if os.path.exists(ph):
    lines += ['## rocprofv3 --kernel-trace --stats  (`python bench.py --headline-only`: only north-star launches)', '',
              '| kernel | calls | avg us | total us |', '|---|---|---|---|']
    for r in csv.DictReader(open(ph)):
        if 'tsamd' in r['Name']:
            lines.append('| `%s` | %s | %.2f | %.1f |' % (short(r['Name'])[:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
            if 'spmm_merge' in r['Name']:
                head_avg = float(r['AverageNs']) / 1e6
    hl = os.path.join(src, 'bench_headline_under_rocprof.log')
    if os.path.exists(hl):
        for line in open(hl):
            if line.startswith('{'):
                hb = json.loads(line)
                lines += ['', 'bench.py line of that profiled run: ms_per_step %.4f, roofline.kernel_ms (HIP events) %.4f; rocprofv3 average of the merge kernel %.4f ms.' % (
                    hb['ms_per_step'], hb['roofline']['kernel_ms'], head_avg or 0.0)]
    lines.append('')
lines += ['## PMC passes (separate runs, `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, 3 SpMM launches each)', '',
          'Units: FETCH_SIZE / WRITE_SIZE are KiB at the L2<->fabric boundary (Infinity-Cache hits included).',
          'gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly 1/2 of the bytes of 16 B/lane',
          'reads, which is what the gather and the (col,value) stream use => fetch bytes = 2 * FETCH_SIZE * 1024.',
          'WRITE_SIZE is uncalibrated on gfx950; it is reported as-is (x1024).', '',
          '| kernel | FETCH_SIZE (KiB, avg) | corrected fetch bytes | WRITE_SIZE (KiB, avg) | write bytes |', '|---|---|---|---|---|']
for k, d in pmc.items():
    f_ = sum(d.get('FETCH_SIZE', [0])) / max(1, len(d.get('FETCH_SIZE', [0])))
    w_ = sum(d.get('WRITE_SIZE', [0])) / max(1, len(d.get('WRITE_SIZE', [0])))
    lines.append('| `%s` | %.0f | %.3e | %.0f | %.3e |' % (k, f_, 2 * f_ * 1024, w_, w_ * 1024))
    if 'spmm_merge_kernel' in k:
        traffic = dict(source='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, %s builder run, scripts/profile_round.sh' % tag, kernel=k, fetch_size_kib=f_, write_size_kib=w_, fetch_bytes_corrected=2 * f_ * 1024,
                       write_bytes=w_ * 1024, hbm_bytes_per_launch=int(2 * f_ * 1024 + w_ * 1024),
                       note='L2<->fabric bytes per launch; FETCH_SIZE doubled per the gfx950 correction, Infinity-Cache hits are included')
if pmc_rel:
    lines += ['', '### the same passes over the relabelled layout (`scripts/prof_spmm_relabelled.py`: no probe, no copy of X)', '',
              '| kernel | FETCH_SIZE (KiB, avg) | corrected fetch bytes | WRITE_SIZE (KiB, avg) | write bytes |', '|---|---|---|---|---|']
    for k, d in pmc_rel.items():
        f_ = sum(d.get('FETCH_SIZE', [0])) / max(1, len(d.get('FETCH_SIZE', [0])))
        w_ = sum(d.get('WRITE_SIZE', [0])) / max(1, len(d.get('WRITE_SIZE', [0])))
        lines.append('| `%s` | %.0f | %.3e | %.0f | %.3e |' % (k, f_, 2 * f_ * 1024, w_, w_ * 1024))
if bench:
    rf = bench['roofline']
    lines += ['', '## bench.py line of the same box', '', '```', json.dumps(bench), '```', '',
              'merge kernel: HIP-event average %.4f ms (bench.py, unprofiled run) vs rocprofv3 average %.4f ms (headline-only trace).' % (
                  rf['kernel_ms'], head_avg if head_avg else float([r for k, r in ts.items() if 'spmm_merge' in k][0]['AverageNs']) / 1e6),
              'algorithmic bytes per launch %.3e -> %.0f GB/s = %.1f %% of the 8 TB/s HBM peak.' % (
                  rf['algorithmic_bytes_per_launch'], rf['achieved'], 100 * rf['frac'])]
    if traffic:
        lines.append('measured fabric traffic per launch %.3e B = %.0f %% of the algorithmic bytes (the rest is L2 reuse of hub columns).' % (
            traffic['hbm_bytes_per_launch'], 100.0 * traffic['hbm_bytes_per_launch'] / rf['algorithmic_bytes_per_launch']))
if bench and bench.get('secondary'):
    lines += ['', '## secondary entries of the same bench.py line (BASELINE configs at full size)', '']
    for e in bench['secondary']:
        lines.append('* `%s`' % json.dumps({k: v for k, v in e.items() if k in ('config', 'has_value', 'ms', 'fw_ms', 'bw_ms', 'gedges_per_s', 'gedges_per_s_fw', 'gproducts_per_s')}) +
                     ' roofline frac %s, parity ok: %s' % (e.get('roofline', {}).get('frac'), e.get('parity', {}).get('ok')))
# secondary kernels (sort / coalesce / spspmm / backward) from the bench_extra trace
pe = os.path.join(src, 'trace_extra', 'extra_kernel_stats.csv')
if os.path.exists(pe):
    lines += ['', '## rocprofv3 --kernel-trace --stats  (`python scripts/bench_extra.py c3 c4 coalesce vbw`)', '',
              '| kernel | calls | avg us | total us |', '|---|---|---|---|']
    for r in csv.DictReader(open(pe)):
        if 'tsamd' in r['Name']:
            lines.append('| `%s` | %s | %.2f | %.1f |' % (short(r['Name'])[:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
for extra in ('bench_c2.json', 'bench_c5.json', 'bench_extra.jsonl'):
    pp = os.path.join(src, extra)
    if os.path.exists(pp):
        import shutil
        shutil.copy(pp, os.path.join(dst, '%s_%s' % (tag, extra)))
open(os.path.join(dst, '%s_summary.md' % tag), 'w').write('\n'.join(lines) + '\n')
if traffic:
    json.dump(traffic, open(os.path.join(dst, 'traffic_%s.json' % workload), 'w'), indent=1)
print('\n'.join(lines))

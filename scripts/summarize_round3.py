"""Condense gpurun_out/<tag>/ (scripts/profile_round3.sh) into profiles/<tag>_summary.md, <tag>_bench_final.json,
<tag>_kernel_stats_headline.csv and refresh profiles/traffic_ns.json from profiles/<tag>_pmc.json.
Usage: python scripts/summarize_round3.py r03 [bench-json-file-name]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
bench_name = sys.argv[2] if len(sys.argv) > 2 else 'bench.json'
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles')


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n.split('(')[0]


# ---- counter traffic of the north-star merge kernel + calibration -> traffic_ns.json ----
d = json.load(open(os.path.join(dst, tag + '_pmc.json')))
w = d['workloads']
ns = w['ns_sum_f32_F128']['kernels']
mk = [k for k in ns if 'spmm_merge_kernel' in k][0]
ctl = w['control_uniform_sum_f32_F128']
ck = [k for k in ctl['kernels'] if 'spmm_merge_kernel' in k][0]
out = dict(source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) + --kernel-trace, round-3 builder run, '
                  'scripts/profile_round3.sh %s -> scripts/profile_pmc.sh %s_pmc (profiles/%s_pmc.{json,md})' % (tag, tag, tag),
           kernel='tsamd::spmm_merge_kernel<float, 4, 0, false, false>',
           fetch_size_kib=ns[mk]['fetch_size_kib_per_call'], write_size_kib=ns[mk]['write_size_kib_per_call'],
           fetch_bytes_corrected=int(2 * ns[mk]['fetch_size_kib_per_call'] * 1024),
           write_bytes=int(ns[mk]['write_size_kib_per_call'] * 1024),
           hbm_bytes_per_launch=ns[mk]['fabric_bytes_per_call_x2_rule'], kernel_us_same_run=ns[mk]['us_per_call'],
           calibration=dict(streaming_copy=d['calibration'], gather_pattern=dict(
               what='the same kernel on the uniform-degree control graph (no hub reuse; X = 4x the Infinity Cache): every '
                    'gathered row is a fabric read, so counter bytes should equal the algorithmic bytes',
               fabric_bytes_x2_rule=ctl['kernels'][ck]['fabric_bytes_per_call_x2_rule'],
               algorithmic_bytes=ctl['info']['algorithmic_bytes'],
               ratio=round(ctl['kernels'][ck]['fabric_bytes_per_call_x2_rule'] / ctl['info']['algorithmic_bytes'], 4))),
           note='L2<->fabric bytes per launch; FETCH_SIZE x 2 (gfx950 rule for 16 B/lane requests) + WRITE_SIZE x 1: both factors '
                'are CONFIRMED on this kernel family -- 2.000 / 1.000 on spmm_permute_rows_kernel (known byte count) and ~1.01 x the '
                'algorithmic bytes on the control graph for the gather pattern itself.  Infinity-Cache hits are included')
json.dump(out, open(os.path.join(dst, 'traffic_ns.json'), 'w'), indent=1)

# ---- bench line, headline-only kernel stats, summary ----
line = open(os.path.join(src, bench_name)).read().strip().split('\n')[-1]
open(os.path.join(dst, tag + '_bench_final.json'), 'w').write(line + '\n')
b = json.loads(line)
b2 = json.loads(open(os.path.join(src, 'bench_headline_under_rocprof.json')).read().strip().split('\n')[-1])
rows = list(csv.DictReader(open(os.path.join(src, 'trace_headline', 'bench_kernel_stats.csv'))))
with open(os.path.join(dst, tag + '_kernel_stats_headline.csv'), 'w') as f:
    wr = csv.writer(f)
    wr.writerow(['kernel', 'calls', 'total_ns', 'avg_ns', 'pct', 'min_ns', 'max_ns'])
    for r in rows:
        wr.writerow([short(r['Name'])[:120], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
ts = [r for r in rows if 'tsamd' in r['Name']]
L = ['# %s profile summary' % tag, '',
     'Source: `scripts/profile_round3.sh %s` on one MI355X (gpurun): (1) `rocprofv3 --kernel-trace --stats -- python bench.py '
     '--headline-only`, (2) plain `python bench.py` (`profiles/%s_bench_final.json`), (3) PMC passes (`profiles/%s_pmc.md`).  '
     'Condensed by `scripts/summarize_round3.py`.' % (tag, tag, tag), '',
     '## rocprofv3 --kernel-trace --stats of `bench.py --headline-only` (every launch of the dominant kernel is a north-star launch)', '',
     '| kernel | calls | avg us | total us |', '|---|---|---|---|']
for r in ts:
    L.append('| `%s` | %s | %.1f | %.1f |' % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
mavg = [float(r['AverageNs']) / 1e3 for r in ts if 'spmm_merge_kernel' in r['Name']][0]
L += ['', 'Dominant kernel `spmm_merge_kernel<float, 4, ADD>`: rocprofv3 average **%.1f us** over the launches above; HIP events on '
      'the launch stream in the same process (`roofline.kernel_ms`): **%.1f us**; in the plain bench run: %.1f us (the kernel\'s '
      'time varies 2.43-2.63 ms between launch series, DESIGN.md 3.1).' % (mavg, b2['roofline']['kernel_ms'] * 1e3, b['roofline']['kernel_ms'] * 1e3),
      '(`spmm_permute_rows_kernel` averages launches that copy X -- the headline steps, operand cache off, ~0.38 ms -- with launches '
      'that return at once because the cached copy still matches -- the `repeated_operand` steps.)', '',
      '## The bench line (`profiles/%s_bench_final.json`)' % tag, '',
      '* headline (operand cache off): %.4f ms/step = %.3f GEdges/s; merge kernel %.4f ms -> `B_alg`/peak %.4f; pre %.4f ms (probe + '
      'copy of X + partition), fix-up %.4f ms; whole op %.4f; counter traffic %.2f GB per launch -> %.4f of the 8 TB/s peak.' % (
          b['ms_per_step'], b['value'], b['roofline']['kernel_ms'], b['roofline']['frac'], b['roofline']['pre_ms'],
          b['roofline']['fixup_ms'], b['roofline']['whole_op_frac'], b['roofline']['traffic'] / 1e9, b['roofline']['frac_traffic']),
      '* the same call again with the same X (operand cache on): %.4f ms/step, bit-identical; relabelled layout API: %.4f ms.' % (
          b['repeated_operand']['ms_per_step'], b['relabelled_layout']['ms_per_step']),
      '* control graph %.4f ms (%.4f); reference CPU kernel on %d cores %.1f ms.' % (
          b['control']['ms'], b['control']['balg_over_peak'], b['cpu_baseline']['cores'], b['cpu_baseline']['ms']),
      '* secondary rows (all `parity.ok` = %s):' % all(s.get('parity', {}).get('ok') for s in b['secondary']), '']
for s in b['secondary']:
    keys = [k for k in s if k.endswith('_ms') or k == 'ms' or k.endswith('_back_to_back')]
    extra = ''
    if s.get('reference_gpu_route', {}).get('ms'):
        extra = '; reference GPU route (hipSPARSE) %.2f ms' % s['reference_gpu_route']['ms']
    L.append('  * `%s`%s: %s%s' % (s['config'], ' (with values)' if s.get('has_value') else '',
                                  ', '.join('%s %s' % (k, json.dumps(s[k])) for k in keys), extra))
open(os.path.join(dst, tag + '_summary.md'), 'w').write('\n'.join(L) + '\n')
print('\n'.join(L))

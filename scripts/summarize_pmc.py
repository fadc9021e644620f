"""Condense gpurun_out/<tag>/ (scripts/profile_pmc.sh over scripts/prof_kernels.py) into
profiles/<tag>.json and profiles/<tag>.md: per workload and kernel the launch count, the average duration of
the un-instrumented trace pass, FETCH_SIZE / WRITE_SIZE per launch, the calibration of both counters on a
kernel whose physical byte count is known exactly, and the fabric bytes per launch they imply."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03_pmc'
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles')


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = n.split('(')[0]
    return n.replace('tsamd::', '')


def plan_of(log):
    for line in open(log):
        if line.startswith('{') and 'workloads' in line:
            return json.loads(line)
    raise SystemExit('no plan line in %s' % log)


def split(rows, name_key, plan):
    """rows in dispatch order -> {label: {kernel: [row, ...]}} using the marker launches."""
    labels = [w['label'] for w in plan['workloads']]
    out, cur = {}, -1
    for r in rows:
        k = short(r[name_key])
        if plan['marker_kernel'] in k:
            cur += 1
            continue
        if cur < 0 or cur >= len(labels) or labels[cur] == '_discard':
            continue
        out.setdefault(labels[cur], {}).setdefault(k, []).append(r)
    return out


plan = plan_of(os.path.join(src, 'trace.log'))
res = {w['label']: dict(info=w, kernels={}) for w in plan['workloads'] if w['label'] != '_discard'}
# ---- durations from the plain kernel trace ------------------------------------------------------------------
tr = [r for r in csv.DictReader(open(os.path.join(src, 'trace', 'k_kernel_trace.csv'))) if 'tsamd' in r['Kernel_Name']]
tr.sort(key=lambda r: int(r['Start_Timestamp']))
for label, ks in split(tr, 'Kernel_Name', plan).items():
    n = res[label]['info']['launches']
    for k, rows in ks.items():
        d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
        res[label]['kernels'][k] = dict(dispatches_per_call=len(rows) / n, us_per_call=sum(d) / n)
# ---- counters ----------------------------------------------------------------------------------------------
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    p = os.path.join(src, 'pmc_%s' % C, 'pmc_counter_collection.csv')
    if not os.path.exists(p):
        continue
    rows = [r for r in csv.DictReader(open(p)) if r['Counter_Name'] == C]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    pl = plan_of(os.path.join(src, 'pmc_%s.log' % C))
    for label, ks in split(rows, 'Kernel_Name', pl).items():
        n = res[label]['info']['launches']
        for k, rs in ks.items():
            res[label]['kernels'].setdefault(k, {})[C.lower() + '_kib_per_call'] = sum(float(r['Counter_Value']) for r in rs) / n

# ---- calibration: spmm_permute_rows_kernel streams B*N*K*s bytes in and out exactly once ---------------------
cal = None
for label, w in res.items():
    c = w['info'].get('calibration')
    if not c:
        continue
    for k, v in w['kernels'].items():
        if c['kernel'] in k and 'fetch_size_kib_per_call' in v:
            cal = dict(kernel=k, workload=label, read_bytes=c['read_bytes'], write_bytes=c['write_bytes'],
                       fetch_factor=c['read_bytes'] / (v['fetch_size_kib_per_call'] * 1024),
                       write_factor=(c['write_bytes'] / (v['write_size_kib_per_call'] * 1024)) if v.get('write_size_kib_per_call') else None)
for label, w in res.items():
    for k, v in w['kernels'].items():
        f, wr = v.get('fetch_size_kib_per_call'), v.get('write_size_kib_per_call')
        if f is not None and wr is not None:
            v['fabric_bytes_per_call_x2_rule'] = int(2 * f * 1024 + wr * 1024)
            if cal and cal['write_factor']:
                v['fabric_bytes_per_call_calibrated'] = int(cal['fetch_factor'] * f * 1024 + cal['write_factor'] * wr * 1024)
            if v.get('us_per_call'):
                v['fabric_tb_per_s_x2_rule'] = round(v['fabric_bytes_per_call_x2_rule'] / v['us_per_call'] / 1e6, 3)
out = dict(source='scripts/profile_pmc.sh %s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) and a plain '
                  '--kernel-trace pass over scripts/prof_kernels.py, one MI355X (gpurun)' % tag,
           units='*_kib_per_call as rocprofv3 prints them (KiB); fabric bytes = L2<->fabric traffic incl. Infinity-Cache hits',
           calibration=cal, workloads=res)
json.dump(out, open(os.path.join(dst, tag + '.json'), 'w'), indent=1)

lines = ['# %s: PMC counters per kernel' % tag, '', out['source'], '']
if cal:
    lines += ['Calibration on `%s` (%s): it reads %d B and writes %d B exactly once per call -> bytes = **%.3f** x FETCH_SIZE, '
              '**%s** x WRITE_SIZE (the guide\'s rule for 16 B/lane streams is 2.0 / uncalibrated).' % (
                  cal['kernel'], cal['workload'], cal['read_bytes'], cal['write_bytes'], cal['fetch_factor'],
                  '%.3f' % cal['write_factor'] if cal['write_factor'] else 'n/a'), '']
lines += ['| workload | kernel | dispatches/call | us/call | FETCH_SIZE KiB | WRITE_SIZE KiB | fabric bytes (2x rule) | priced against (SURVEY 8d) | fabric / priced |',
          '|---|---|---|---|---|---|---|---|---|']
for label, w in res.items():
    ks = sorted(w['kernels'].items(), key=lambda kv: -kv[1].get('us_per_call', 0))
    ab = w['info'].get('algorithmic_bytes')
    for i, (k, v) in enumerate(ks):
        fb = v.get('fabric_bytes_per_call_x2_rule')
        lines.append('| %s | `%s` | %.1f | %.1f | %s | %s | %s | %s | %s |' % (
            label if i == 0 else '', k[:70], v.get('dispatches_per_call', 0), v.get('us_per_call', 0),
            '%.0f' % v['fetch_size_kib_per_call'] if 'fetch_size_kib_per_call' in v else '-',
            '%.0f' % v['write_size_kib_per_call'] if 'write_size_kib_per_call' in v else '-',
            fb if fb is not None else '-', ab if (i == 0 and ab) else '', ('%.2f' % (fb / ab)) if (fb and ab and i == 0) else ''))
open(os.path.join(dst, tag + '.md'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))

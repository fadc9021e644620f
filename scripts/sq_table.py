"""rocprofv3 --pmc SQ_* counter CSV -> markdown table per kernel (the issue-side view of profiles/r0N_sq_counters.md).
Usage: python scripts/sq_table.py counter_collection.csv [more.csv ...]
VALU instr / wave = SQ_INSTS_VALU / SQ_WAVES; VALU pipe = waves x VALU instr x 4 cycles / (1024 SIMDs x time x 2.4 GHz);
active / VALU / waiting / issue stall = SQ_ACTIVE_INST_ANY, SQ_ACTIVE_INST_VALU, SQ_WAIT_ANY, SQ_WAIT_INST_ANY over
SQ_WAVE_CYCLES (the counters are summed over the shader engines rocprofv3 reports)."""
import collections
import csv
import re
import sys


def short(name):
    name = name.split('(')[0] if not name.startswith('void') else name[5:].split('(')[0]
    return re.sub(r'tsamd::|\(anonymous namespace\)::', '', name)[:70]


agg = collections.OrderedDict()
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        if 'tsamd' not in r['Kernel_Name']:
            continue
        k = short(r['Kernel_Name'].replace('(anonymous namespace)::', ''))
        a = agg.setdefault(k, dict(c=collections.defaultdict(float), d={}, ))
        a['c'][r['Counter_Name']] += float(r['Counter_Value'])
        a['d'][(path, r['Dispatch_Id'])] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print('| kernel | launches | waves | us | VALU instr / wave | VALU pipe | active | VALU | waiting | issue stall |')
print('|---|---|---|---|---|---|---|---|---|---|')
for k, a in agg.items():
    n = len(a['d'])
    c = {kk: v / n for kk, v in a['c'].items()}
    us = sum(a['d'].values()) / n
    waves = c.get('SQ_WAVES', 0.0)
    if waves <= 0 or us <= 0:
        continue
    ipw = c.get('SQ_INSTS_VALU', 0.0) / waves
    pipe = waves * ipw * 4.0 / (1024.0 * us * 1e-6 * 2.4e9)
    wc = max(c.get('SQ_WAVE_CYCLES', 0.0), 1.0)
    print('| `%s` | %d | %d | %.1f | %.0f | %.2f | %.2f | %.2f | %.2f | %.2f |' % (
        k, n, waves, us, ipw, pipe, c.get('SQ_ACTIVE_INST_ANY', 0) / wc, c.get('SQ_ACTIVE_INST_VALU', 0) / wc,
        c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc))

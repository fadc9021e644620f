"""Condense a rocprofv3 *_kernel_stats.csv: short kernel names, tsamd kernels first."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
only = len(sys.argv) > 2 and sys.argv[2] == '--tsamd'
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    m = re.match(r'(?:void )?([\w:]+(?:<[^(]{0,60})?)', n)
    return (m.group(1) if m else n)[:90]
print('%-92s %6s %12s %12s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
for r in rows:
    name = r['Name']
    if only and 'tsamd' not in name: continue
    print('%-92s %6s %12.1f %12.2f %7s' % (short(name), r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3, r['Percentage']))

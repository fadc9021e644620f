"""Print the kernel timeline of the LAST `n` dispatches of a rocprofv3 kernel_trace.csv: start offset, duration, gap."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-n:]
t0 = int(rows[0]['Start_Timestamp'])
prev_end = t0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = re.sub(r'^void ', '', name)[:70]
    print('%9.1f us  +%7.1f  gap %6.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name))
    prev_end = e

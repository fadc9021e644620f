"""Heuristic scan of hipcc --save-temps assembly: kernels in which a global load is waited for (s_waitcnt vmcnt(0)) with no
other load in flight, several times in a row -- the shape of the two latency bugs found in round 5 (masked SDDMM: record word,
wait, gathers, wait; winner records on int32 ids: four loads, four waits).  Prints kernels by the number of
"lone load -> full wait" events inside loops.  Usage: python scripts/scan_serial_loads.py file.s [...]"""
import re
import subprocess
import sys

for path in sys.argv[1:]:
    txt = open(path).read()
    funcs = re.split(r'\n(?=_Z[^\n]*:\s*;? ?@?)', txt)
    rows = []
    for f in funcs:
        name = f.split(':', 1)[0].strip()
        if not name.startswith('_Z') or 'kernel' not in name:
            continue
        lines = f.split('\n')
        inflight, lone_waits, waits, loads = 0, 0, 0, 0
        for ln in lines:
            t = ln.strip()
            if t.startswith('global_load') or t.startswith('buffer_load'):
                inflight += 1
                loads += 1
            elif t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
                waits += 1
                if inflight == 1:
                    lone_waits += 1
                inflight = 0
            elif t.startswith('.LBB') or t.startswith('s_cbranch') or t.startswith('s_branch'):
                pass
        if loads:
            dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r'\(anonymous namespace\)::', '', dem).split('(')[0].replace('void tsamd::', '')
            rows.append((lone_waits, waits, loads, dem[:90]))
    rows.sort(reverse=True)
    print('==', path.split('/')[-1])
    for r in rows[:14]:
        print('  lone-load waits %3d  full waits %3d  loads %3d  %s' % r)

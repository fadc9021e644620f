"""Register pressure of every kernel of the library: compiles each csrc/*.hip with `hipcc --save-temps` (gfx950) into a
scratch directory and lists, from the code objects' metadata, the kernels that spill VGPRs / use scratch memory and the
ones with the most SGPR spills.  CPU only (cross-compile), ~2 min on 8 cores.

    python scripts/scan_spills.py [/tmp/scan_spills] [--sgpr 48]

Round 6 found the compacting bucket sort at 128 VGPRs + 16..61 spilled this way (profiles/r06_ab_coalesce_fused.md)."""
import glob
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'pytorch_sparse_amd', 'csrc')
out = next((a for a in sys.argv[1:] if not a.startswith('--')), '/tmp/scan_spills')
sgpr_min = int(sys.argv[sys.argv.index('--sgpr') + 1]) if '--sgpr' in sys.argv else 48


def build(src):
    d = os.path.join(out, os.path.basename(src)[:-4])
    os.makedirs(d, exist_ok=True)
    subprocess.run(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC,
                    '-c', src, '--save-temps', '-o', 'x.o'], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return d


with ThreadPoolExecutor(8) as ex:
    dirs = list(ex.map(build, sorted(glob.glob(os.path.join(CSRC, '*.hip')))))
rows = []
for d in dirs:
    for f in glob.glob(os.path.join(d, '*gfx950.s')):
        s = open(f).read()
        for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', s, re.S):
            body = m.group(2)

            def g(k):
                mm = re.search(k + r':\s+(\d+)', body)
                return int(mm.group(1)) if mm else -1
            name = re.sub(r'^_ZN5tsamd12_GLOBAL__N_1\d+', '', m.group(1))
            rows.append((os.path.basename(d), name[:100], g(r'\.vgpr_count'), g(r'\.vgpr_spill_count'), g(r'\.private_segment_fixed_size'),
                         g(r'\.sgpr_spill_count'), g(r'\.group_segment_fixed_size')))
print('%d kernels' % len(rows))
print('-- VGPR spills / scratch:')
for r in sorted(rows):
    if r[3] > 0 or r[4] > 0:
        print('  %-14s %-100s vgpr %3d spilled %3d scratch %4d B' % r[:5])
print('-- SGPR spills >= %d:' % sgpr_min)
for r in sorted(rows, key=lambda r: -r[5]):
    if r[5] >= sgpr_min:
        print('  %-14s %-100s vgpr %3d sgpr spills %3d' % (r[0], r[1], r[2], r[5]))

"""Per-kernel VGPR / occupancy table from a `hipcc -Rpass-analysis=kernel-resource-usage` log:
    python scripts/kres.py log.txt [substring]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0].strip().split(' ')[0]
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r'\(anonymous namespace\)::', '', dem).split('(')[0].replace('void tsamd::', '')
    if pat and pat not in dem:
        continue
    g = lambda k: re.search(k + r': (\d+)', b).group(1)
    print('%-70s VGPR %3s SGPR %3s occ %s scratch %s LDS %s' % (dem[:70], g('VGPRs'), g('SGPRs'), g(r'Occupancy \[waves/SIMD\]'),
                                                              g(r'ScratchSize \[bytes/lane\]'), g(r'LDS Size \[bytes/block\]')))

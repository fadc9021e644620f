#!/bin/bash
# sort tiles of the 12-byte modes (pairs / riding values): 24 (shipped) vs 16 / 20 / 28 entries per thread
mkdir -p gpurun_out/r5ar
R=$PWD
for v in shipped pairs16 pairs20 pairs28 shipped; do
  echo "== $v" >> gpurun_out/r5ar/ab.log
  if [ $v = shipped ]; then unset LD_PRELOAD; else export LD_PRELOAD=$R/build/ab/$v.so; fi
  python scripts/bench_sort.py --big 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5ar/ab.log
done
unset LD_PRELOAD
python - <<'P'
import json
for l in open('gpurun_out/r5ar/ab.log'):
    if l.startswith('=='): print(l.strip()); continue
    try: d = json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print({k: d[k] for k in ('E', 'sort_coo_ms', 'construct_ms', 'coalesce_ms', 't_ms')})
P

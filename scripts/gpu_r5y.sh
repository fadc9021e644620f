#!/bin/bash
# select / filter kernels with the predicate dispatched once per tile (loads issued together) against the old ones
mkdir -p gpurun_out/r5y
R=$PWD
python -m pytest tests/test_select_gpu.py tests/test_api_gpu.py tests/test_random_cases_gpu.py -q -m gpu -x 2>&1 | tail -2
for v in shipped old shipped old; do
  echo "== $v" >> gpurun_out/r5y/ab.log
  if [ $v = old ]; then export LD_PRELOAD=$R/build/ab/oldselect.so; else unset LD_PRELOAD; fi
  python scripts/bench_select.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5y/ab.log
done
unset LD_PRELOAD
python - <<'P'
import json
for l in open('gpurun_out/r5y/ab.log'):
    if l.startswith('=='): print(l.strip()); continue
    try: d = json.loads(l)
    except Exception: print(l.strip()[:160]); continue
    print({k: d[k] for k in d if k in ('op', 'name', 'bench', 'ms', 'ref_ms', 'gbs')})
P

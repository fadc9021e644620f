#!/bin/bash
# neighbour exchange of the order probe / range check / head flags on DPP (shipped) against ds_bpermute (olddpp.so)
mkdir -p gpurun_out/r5ac
R=$PWD
python -m pytest tests/test_sort_gpu.py tests/test_api_gpu.py tests/test_random_cases_gpu.py -q -m gpu -x 2>&1 | tail -2
for v in shipped old shipped old; do
  echo "== $v" >> gpurun_out/r5ac/ab.log
  if [ $v = old ]; then export LD_PRELOAD=$R/build/ab/olddpp.so; else unset LD_PRELOAD; fi
  python scripts/bench_sort.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5ac/ab.log
done
unset LD_PRELOAD
python - <<'P'
import json
for l in open('gpurun_out/r5ac/ab.log'):
    if l.startswith('=='): print(l.strip()); continue
    try: d = json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print({k: d[k] for k in ('sort_coo_ms', 'sort_coo_auto_ms', 'coalesce_index_ms', 'coo_check_ms', 'construct_ms', 'coalesce_ms', 'transpose_ms', 't_ms')})
P

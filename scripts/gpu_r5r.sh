#!/bin/bash
# round 5: values ride the sort in the functional coalesce / transpose too
mkdir -p gpurun_out/r5r
( python -m pytest tests/test_sort_gpu.py tests/test_api_gpu.py tests/test_random_cases_gpu.py tests/test_jit.py -x -q -m gpu ) > gpurun_out/r5r/pytest.log 2>&1
tail -3 gpurun_out/r5r/pytest.log
for v in 1 0; do
  echo "== ride=$v" >> gpurun_out/r5r/ab.log
  TSAMD_SORT_VALUE_RIDE=$v python scripts/bench_sort.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5r/ab.log
done
python - <<'P'
import json
for l in open('gpurun_out/r5r/ab.log'):
    if l.startswith('=='): print(l.strip()); continue
    try: d = json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print({k: d[k] for k in ('E', 'sort_coo_ms', 'construct_ms', 'coalesce_ms', 'transpose_ms', 't_ms', 'construct_frac_of_peak')})
P

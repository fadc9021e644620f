#!/bin/bash
# pull vs scatter with grad_value, by row size (rule of storage_spmm), after the pipelined masked SDDMM
mkdir -p gpurun_out/r5o
( python -m pytest tests/test_spmm_gpu.py -x -q -k "minmax or masked" ) > gpurun_out/r5o/pytest.log 2>&1
tail -3 gpurun_out/r5o/pytest.log
for cfg in "bf16 32" "bf16 64" "bf16 256" "f16 64" "f32 32" "f32 64" "f32 256" ; do
  set -- $cfg
  echo "== $1 K=$2" >> gpurun_out/r5o/ab.log
  DTYPE=$1 K=$2 python scripts/bench_minmax_bw.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5o/ab.log
done
python - <<'P'
import json
for l in open('gpurun_out/r5o/ab.log'):
    if l.startswith('=='): print(l.strip()); continue
    try: d = json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print({k: d[k] for k in ('has_value', 'masks_mat_ms', 'masks_mat_value_ms', 'scatter_mat_ms', 'scatter_mat_value_ms') if k in d})
P
python scripts/ab_arg32.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5o/arg32.log; python - <<'P'
import json
for l in open('gpurun_out/r5o/arg32.log'):
    try: d = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print({k: d[k] for k in d if not k.endswith('_b_fw_ms') })
P

#!/bin/bash
# round 5, second session: pipelined masked SDDMM + int32 winner ids -- tests, then same-box A/B
mkdir -p gpurun_out/r5n
( python -m pytest tests/test_spmm_gpu.py -x -q -k "minmax or masked" ) > gpurun_out/r5n/pytest.log 2>&1
tail -5 gpurun_out/r5n/pytest.log
for v in pipe nopipe pipe nopipe; do
  if [ $v = pipe ]; then unset TSAMD_MASKED_SDDMM_PIPE; else export TSAMD_MASKED_SDDMM_PIPE=0; fi
  for dt in bf16 f32; do
    echo "== $v $dt" >> gpurun_out/r5n/ab.log
    DTYPE=$dt python scripts/bench_minmax_bw.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5n/ab.log
  done
done
unset TSAMD_MASKED_SDDMM_PIPE
python scripts/ab_arg32.py > gpurun_out/r5n/arg32.log 2>&1
python - <<'P'
import json
for l in open('gpurun_out/r5n/ab.log'):
    if l.startswith('=='): print(l.strip()); continue
    try: d = json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print({k: d[k] for k in ('has_value', 'masks_mat_ms', 'masks_mat_value_ms', 'scatter_mat_value_ms') if k in d})
P
cat gpurun_out/r5n/arg32.log | grep -v amdgpu.ids

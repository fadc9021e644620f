#!/bin/bash
mkdir -p gpurun_out/r5x
python -m pytest tests/test_spmm_gpu.py -q -m gpu -x -k "minmax or masked or route" 2>&1 | tail -2
python scripts/ab_winrec_arg32.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5x/ab.jsonl
python scripts/ab_arg32.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5x/arg32.jsonl
python - <<'P'
import json
for l in open('gpurun_out/r5x/arg32.jsonl'):
    try: d = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print({k: d[k] for k in d if k.endswith('fwbw_ms') or k in ('dtype','K','has_value','same_bits','arg32_fw_ms','arg64_fw_ms')})
P

#!/bin/bash
mkdir -p gpurun_out/r5ah
python -m pytest tests/test_api_gpu.py tests/test_configs_gpu.py tests/test_random_cases_gpu.py -q -m gpu -k "spspmm or random" 2>&1 | tail -2
for rep in 1 2 3; do VARIANT=shipped python scripts/ab_spspmm_r5.py stress c4 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r5ah/ab.jsonl; done
cat gpurun_out/r5ah/ab.jsonl

#!/bin/bash
mkdir -p gpurun_out/r5w
python scripts/ab_winrec_arg32.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5w/ab.jsonl

#!/bin/bash
# winner records: rows per step (loads in flight) 2 / 4 (shipped) / 8, same box
mkdir -p gpurun_out/r5ab
R=$PWD
python -m pytest tests/test_spmm_gpu.py -q -m gpu -x -k "minmax or masked or route" 2>&1 | tail -2
for rep in 1 2; do
for v in shipped rows2 rows8; do
  if [ $v = shipped ]; then unset TSAMD_LIB; else export TSAMD_LIB=$R/build/ab/$v.so; fi
  echo "== $v" >> gpurun_out/r5ab/ab.log
  python scripts/ab_winrec_arg32.py 2>&1 | grep -v amdgpu.ids | head -2 >> gpurun_out/r5ab/ab.log
done; done
cat gpurun_out/r5ab/ab.log

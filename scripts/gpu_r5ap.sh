#!/bin/bash
# winner records: full-line stores (shipped) vs two 16-byte stores per lane (noline)
mkdir -p gpurun_out/r5ap
R=$PWD
python -m pytest tests/test_spmm_gpu.py tests/test_configs_gpu.py -q -m gpu -x -k "minmax or masked or route or c3" 2>&1 | tail -2
for rep in 1 2; do
for v in shipped noline; do
  if [ $v = shipped ]; then unset TSAMD_LIB; else export TSAMD_LIB=$R/build/ab/$v.so; fi
  echo "== $v" >> gpurun_out/r5ap/ab.log
  python scripts/ab_winrec_arg32.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5ap/ab.log
done; done
cat gpurun_out/r5ap/ab.log

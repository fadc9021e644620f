#!/bin/bash
# round 5, second session: masked SDDMM with v_dot2c (shipped) against the per-element form (build/ab/nodot2.so)
mkdir -p gpurun_out/r5m
( python -m pytest tests/test_spmm_gpu.py -x -q -k "minmax_bw" ) > gpurun_out/r5m/pytest.log 2>&1
tail -3 gpurun_out/r5m/pytest.log
for v in shipped nodot2 shipped nodot2; do
  if [ $v = shipped ]; then unset TSAMD_LIB; else export TSAMD_LIB=$PWD/build/ab/$v.so; fi
  for dt in bf16 f16; do
    echo "== $v $dt" >> gpurun_out/r5m/ab.log
    DTYPE=$dt python scripts/bench_minmax_bw.py >> gpurun_out/r5m/ab.log 2>&1
  done
done
cat gpurun_out/r5m/ab.log | cut -c1-600

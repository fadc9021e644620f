#!/bin/bash
# Round 4: masked SDDMM (grad_value of the min / max pull backward) skips slots without winners -- parity + timing.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04g; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_spmm_gpu.py -x -q -m gpu -k "minmax_bw or value_bw" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
K=128 DTYPE=bf16 timeout 100 python scripts/bench_minmax_bw.py > $OUT/minmax_bw_bf16.log 2>&1
grep -h masks_mat_ms $OUT/minmax_bw_bf16.log | sed 's/"lists_sum.*//' | cut -c1-400

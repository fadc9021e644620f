#!/bin/bash
# Round 4, third GPU call: the masked sum of the min / max pull backward skips segments without winners -- parity,
# same-box A/B against build/ab/r3loops.so (whole rows gathered), SQ counter pass over the kernels touched this round.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04c}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_spmm_gpu.py -x -q -m gpu -k "minmax or bw or masked" ) > $OUT/pytest_a.log 2>&1
tail -3 $OUT/pytest_a.log
( time timeout 600 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "c3" ) > $OUT/pytest_b.log 2>&1
tail -3 $OUT/pytest_b.log
cp pytorch_sparse_amd/lib/libtsamd.so build/ab/new.so
for cfg in "128 bf16" "256 bf16" "64 bf16" "128 f32" "32 f32"; do
  set -- $cfg
  for lib in r3loops new; do
    echo "== K=$1 $2 $lib" >> $OUT/ab_minmax_bw.log
    K=$1 DTYPE=$2 TSAMD_LIB=$GRAFT_REPO_ROOT/build/ab/$lib.so timeout 200 python scripts/bench_minmax_bw.py >> $OUT/ab_minmax_bw.log 2>&1
  done
done
grep -E "==|masks_mat_ms" $OUT/ab_minmax_bw.log | sed 's/"lists_sum.*//' | cut -c1-260
bash scripts/profile_sq.sh ${1:-r04c} c3_max_fw_bf16_F128 c3_max_bw_pull_bf16_F128 c3_max_bw_pull_val_bf16_F128 c4_spspmm sort_coo_7m5 coalesce_7m5

#!/bin/bash
# Round 4, fourth GPU call: persistent, row-pipelined small-row SpSpMM kernels (parity + same-box A/B against
# build/ab/nopipe.so = -DTSAMD_SPSPMM_ROW_PIPE=0), wave-uniform skip of empty slots in the masked sum.
set -u
TAG=${1:-r04d}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_api_gpu.py tests/test_random_cases_gpu.py tests/test_jit.py -x -q -m gpu ) > $OUT/pytest_a.log 2>&1
tail -3 $OUT/pytest_a.log
( time timeout 600 python -m pytest tests/test_configs_gpu.py tests/test_spmm_gpu.py -x -q -m gpu -k "c4 or stress or c3 or minmax or bw" ) > $OUT/pytest_b.log 2>&1
tail -3 $OUT/pytest_b.log
for i in 1 2; do
LD_PRELOAD=$GRAFT_REPO_ROOT/build/ab/nopipe.so timeout 300 python scripts/prof_spspmm.py c4 stress >> $OUT/spspmm_nopipe.log 2>&1
timeout 300 python scripts/prof_spspmm.py c4 stress >> $OUT/spspmm_pipe.log 2>&1
done
echo "nopipe:"; grep -h '"ms"' $OUT/spspmm_nopipe.log | cut -c1-30,150-200
echo "pipe:"; grep -h '"ms"' $OUT/spspmm_pipe.log | cut -c1-30,150-200
K=128 DTYPE=bf16 timeout 200 python scripts/bench_minmax_bw.py > $OUT/minmax_bw_bf16.log 2>&1
K=128 DTYPE=f32 timeout 200 python scripts/bench_minmax_bw.py > $OUT/minmax_bw_f32.log 2>&1
grep -h masks_mat_ms $OUT/minmax_bw_*.log | sed 's/"lists_sum.*//' | cut -c1-300
bash scripts/profile_sq.sh $TAG c4_spspmm c3_max_bw_pull_bf16_F128 > $OUT/sq_stdout.log 2>&1
grep -E "spspmm_symbolic_small|numeric_small_pipe|true>" $OUT/sq_stdout.log

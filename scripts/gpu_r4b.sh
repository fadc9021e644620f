#!/bin/bash
# Round 4, second GPU call: correctness of the reworked min / max step loop and the one-wave SpSpMM expansion, same-box
# A/B against the round-3 loops (build/ab/r3loops.so), gloo rehearsal of the self-launching `bench.py --gpus 2`.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04b}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_spmm_gpu.py tests/test_api_gpu.py -x -q -m gpu ) > $OUT/pytest_a.log 2>&1
tail -3 $OUT/pytest_a.log
( time timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_random_cases_gpu.py -x -q -m gpu -k "c3 or c4 or stress or random" ) > $OUT/pytest_b.log 2>&1
tail -3 $OUT/pytest_b.log
cp pytorch_sparse_amd/lib/libtsamd.so build/ab/new.so
timeout 300 python scripts/ab_minmax_fw.py r3loops new r3loops new > $OUT/ab_minmax_fw.log 2>&1
cat $OUT/ab_minmax_fw.log
LD_PRELOAD=$GRAFT_REPO_ROOT/build/ab/r3loops.so timeout 300 python scripts/prof_spspmm.py c4 stress > $OUT/spspmm_r3loops.log 2>&1
timeout 300 python scripts/prof_spspmm.py c4 stress > $OUT/spspmm_new.log 2>&1
LD_PRELOAD=$GRAFT_REPO_ROOT/build/ab/r3loops.so timeout 300 python scripts/prof_spspmm.py c4 stress >> $OUT/spspmm_r3loops.log 2>&1
timeout 300 python scripts/prof_spspmm.py c4 stress >> $OUT/spspmm_new.log 2>&1
grep -h '"ms"' $OUT/spspmm_r3loops.log | cut -c1-200
echo ---
grep -h '"ms"' $OUT/spspmm_new.log | cut -c1-200
( time timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 > $OUT/bench_n2.json 2> $OUT/bench_n2.err ) 2> $OUT/bench_n2.time
tail -2 $OUT/bench_n2.time; cut -c1-600 $OUT/bench_n2.json; tail -5 $OUT/bench_n2.err

#!/bin/bash
# round 6 (re-entry): the three differential fuzzers with fresh seeds on the final kernels
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6ag
mkdir -p $OUT
timeout 420 python tests/fuzz_index_ops.py --cases 300 --seed 61 > $OUT/fuzz_index_ops.log 2>&1; echo "index rc=$?" | tee -a $OUT/summary.txt
timeout 420 python tests/fuzz_spmm.py --cases 500 --seed 62 > $OUT/fuzz_spmm.log 2>&1; echo "spmm rc=$?" | tee -a $OUT/summary.txt
timeout 200 python tests/fuzz_records.py 150 63 > $OUT/fuzz_records.log 2>&1; echo "records rc=$?" | tee -a $OUT/summary.txt
tail -2 $OUT/fuzz_index_ops.log $OUT/fuzz_spmm.log $OUT/fuzz_records.log

#!/bin/bash
# round 6 (re-entry): medium-row SpSpMM kernels forked onto a side stream: tests, same-box A/B (configs[3], stress)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6af
mkdir -p $OUT
python -m pytest tests/test_api_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "spspmm or c4" 2>&1 | tail -3
for rep in 1 2 3; do
  VARIANT=fork python scripts/ab_spspmm_r5.py c4 stress >> $OUT/ab.jsonl 2>> $OUT/ab.err
  VARIANT=nofork LD_PRELOAD=build/ab/nofork.so python scripts/ab_spspmm_r5.py c4 stress >> $OUT/ab.jsonl 2>> $OUT/ab.err
done
cat $OUT/ab.jsonl; tail -2 $OUT/ab.err

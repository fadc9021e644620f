#!/bin/bash
# Round 4: SpSpMM small-row micro-steps (one LDS read per product, 24-bit hash multiply, packet stores of the keys):
# parity + same-box A/B against build/ab/nopipe.so (the library before them).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04h; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_api_gpu.py -x -q -m gpu -k "spspmm or fuzz" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for i in 1 2; do
LD_PRELOAD=$GRAFT_REPO_ROOT/build/ab/nopipe.so timeout 100 python scripts/prof_spspmm.py c4 >> $OUT/spspmm_before.log 2>&1
timeout 100 python scripts/prof_spspmm.py c4 >> $OUT/spspmm_after.log 2>&1
done
echo before; grep -h '"ms"' $OUT/spspmm_before.log | sed 's/.*"ms": \([0-9.]*\).*/\1/'
echo after; grep -h '"ms"' $OUT/spspmm_after.log | sed 's/.*"ms": \([0-9.]*\).*/\1/'

#!/bin/bash
# Round 4, last call: the final tree -- smoke(), the full-size config tests the last changes touch, SpSpMM API tests.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04i; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log
( time timeout 400 python -m pytest tests/test_configs_gpu.py tests/test_api_gpu.py -x -q -m gpu -k "c3 or c4 or c5 or spspmm" ) > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log

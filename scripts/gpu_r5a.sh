#!/bin/bash
# round 5, call A: full GPU suite with durations + default bench (new compact line)
mkdir -p gpurun_out/r5a
( time python -m pytest tests -m gpu -x -q --durations=40 ) > gpurun_out/r5a/pytest.log 2>&1
tail -60 gpurun_out/r5a/pytest.log
( time python bench.py ) > gpurun_out/r5a/bench.out 2> gpurun_out/r5a/bench.err
echo "bench rc $?"
wc -c gpurun_out/r5a/bench.out
cat gpurun_out/r5a/bench.out
grep "^\[bench\] secondary\|^real" gpurun_out/r5a/bench.err

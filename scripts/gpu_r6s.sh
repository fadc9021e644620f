#!/bin/bash
# round 6, session 3: GPU suite + default bench line on the tree as found
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r6s}
mkdir -p gpurun_out/$T
( time python -m pytest tests -m gpu -x -q --durations=10 ) > gpurun_out/$T/pytest_full.log 2>&1
tail -4 gpurun_out/$T/pytest_full.log
( time python bench.py ) > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
cp profiles/bench_last_full.json gpurun_out/$T/bench_full.json 2>/dev/null
tail -c 3500 gpurun_out/$T/bench.json
tail -3 gpurun_out/$T/bench.err | cut -c1-300

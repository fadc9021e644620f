#!/bin/bash
# full GPU suite on the current tree (timing + log for profiles/)
mkdir -p gpurun_out/r5s
( time python -m pytest tests -m gpu -x -q --durations=25 ) > gpurun_out/r5s/pytest_full.log 2>&1
tail -45 gpurun_out/r5s/pytest_full.log

#!/bin/bash
mkdir -p gpurun_out/r5l
( time python -m pytest tests -m gpu -x -q --durations=25 ) > gpurun_out/r5l/pytest.log 2>&1
tail -45 gpurun_out/r5l/pytest.log

#!/bin/bash
# round 5, final evidence on the final tree: full GPU suite, headline under rocprofv3, default bench, stress row,
# PMC table, N = 2 gloo rehearsal of bench.py --gpus 2
mkdir -p gpurun_out/r5z
( time python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/r5z/pytest_full.log 2>&1
tail -4 gpurun_out/r5z/pytest_full.log
bash scripts/profile_round5.sh r05_final > gpurun_out/r5z/profile_round5.log 2>&1
tail -6 gpurun_out/r5z/profile_round5.log | cut -c1-300
bash scripts/profile_pmc.sh r05c_pmc > gpurun_out/r5z/pmc.log 2>&1
tail -1 gpurun_out/r5z/pmc.log
( time TSAMD_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 ) > gpurun_out/r5z/bench_n2_gloo.json 2> gpurun_out/r5z/bench_n2_gloo.err
tail -c 1500 gpurun_out/r5z/bench_n2_gloo.json; tail -3 gpurun_out/r5z/bench_n2_gloo.err | cut -c1-200

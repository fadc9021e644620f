#!/bin/bash
# round 6 evidence on one box: full GPU suite, headline under rocprofv3, default bench, stress row, PMC table,
# N = 2 gloo rehearsals of bench.py --gpus 2 (weak: configs[4] share; strong: the north-star matrix)
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r6z}
mkdir -p gpurun_out/$T
( time python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/$T/pytest_full.log 2>&1
tail -4 gpurun_out/$T/pytest_full.log
bash scripts/profile_round6.sh r06_final > gpurun_out/$T/profile_round6.log 2>&1
tail -6 gpurun_out/$T/profile_round6.log | cut -c1-300
bash scripts/profile_pmc.sh r06_pmc > gpurun_out/$T/pmc.log 2>&1
tail -1 gpurun_out/$T/pmc.log
( time TSAMD_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 ) > gpurun_out/$T/bench_n2_gloo.json 2> gpurun_out/$T/bench_n2_gloo.err
tail -c 1200 gpurun_out/$T/bench_n2_gloo.json; tail -3 gpurun_out/$T/bench_n2_gloo.err | cut -c1-200
( time TSAMD_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --workload ns --scaling strong ) > gpurun_out/$T/bench_n2_gloo_strong.json 2> gpurun_out/$T/bench_n2_gloo_strong.err
tail -c 1200 gpurun_out/$T/bench_n2_gloo_strong.json; tail -3 gpurun_out/$T/bench_n2_gloo_strong.err | cut -c1-200

#!/bin/bash
# round 5, second session: the suites touched by the arg32 / route-rule / SDDMM changes + the default bench run
mkdir -p gpurun_out/r5p
( time python -m pytest tests/test_spmm_gpu.py tests/test_jit.py tests/test_api_gpu.py tests/test_operand_cache_gpu.py tests/test_parallel_gpu.py tests/test_sort_gpu.py tests/test_cabi.py -x -q -m gpu --durations=8 ) > gpurun_out/r5p/pytest_a.log 2>&1
tail -14 gpurun_out/r5p/pytest_a.log
( time python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "c3" --durations=5 ) > gpurun_out/r5p/pytest_c3.log 2>&1
tail -8 gpurun_out/r5p/pytest_c3.log
( time python bench.py ) > gpurun_out/r5p/bench_stdout.log 2> gpurun_out/r5p/bench_stderr.log
tail -c 3200 gpurun_out/r5p/bench_stdout.log
cp profiles/bench_last_full.json gpurun_out/r5p/bench_full.json 2>/dev/null
tail -4 gpurun_out/r5p/bench_stderr.log | cut -c1-300

#!/bin/bash
# round 6: device-driven hetero samplers -- tests, then the sampler bench with the new and the round-5 operator library
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6t
python -m pytest tests/test_hetero_sample_gpu.py tests/test_sample_gpu.py -m gpu -x -q 2>&1 | tail -30
python scripts/bench_sample.py > gpurun_out/r6t/bench_sample_new.jsonl 2> gpurun_out/r6t/bench_sample_new.err
grep "neighbor_sample" gpurun_out/r6t/bench_sample_new.jsonl
tail -3 gpurun_out/r6t/bench_sample_new.err
cp build/ab/_tsamd_ops_r5sampler.so pytorch_sparse_amd/lib/_tsamd_ops.so
python scripts/bench_sample.py > gpurun_out/r6t/bench_sample_r5ops.jsonl 2> gpurun_out/r6t/bench_sample_r5ops.err
grep "neighbor_sample" gpurun_out/r6t/bench_sample_r5ops.jsonl
tail -3 gpurun_out/r6t/bench_sample_r5ops.err

#!/bin/bash
# smoke() as the driver runs it + the bench with the stress row
mkdir -p gpurun_out/r5an
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | tail -5
( time python bench.py --stress ) > gpurun_out/r5an/bench_stress.json 2> gpurun_out/r5an/bench_stress.err
tail -c 700 gpurun_out/r5an/bench_stress.json; echo; tail -3 gpurun_out/r5an/bench_stress.err | cut -c1-200

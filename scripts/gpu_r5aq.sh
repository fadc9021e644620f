#!/bin/bash
# last check of the final tree: full GPU suite + default bench
mkdir -p gpurun_out/r5aq
( time python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/r5aq/pytest_full.log 2>&1
tail -4 gpurun_out/r5aq/pytest_full.log
( time python bench.py ) > gpurun_out/r5aq/bench.json 2> gpurun_out/r5aq/bench.err
wc -c gpurun_out/r5aq/bench.json; cp profiles/bench_last_full.json gpurun_out/r5aq/bench_full.json
python - <<'P'
import json
d=json.loads(open('gpurun_out/r5aq/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['wall_s'], all(r.get('ok') for r in d['secondary']), d['parity']['ok'])
P

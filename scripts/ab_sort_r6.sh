#!/bin/bash
# same-box A/B of sort variants: build/ab/*.so in front of the shipped library.  usage: ab_sort_r6.sh out.jsonl [flags]
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$1; shift
mkdir -p gpurun_out
: > gpurun_out/$OUT
for rep in 1 2; do
  python scripts/bench_sort_quick.py shipped "$@" >> gpurun_out/$OUT 2>> gpurun_out/$OUT.err
  for so in build/ab/*.so; do
    [ -f "$so" ] || continue
    LD_PRELOAD=$so python scripts/bench_sort_quick.py $(basename $so .so) "$@" >> gpurun_out/$OUT 2>> gpurun_out/$OUT.err
  done
done
cat gpurun_out/$OUT

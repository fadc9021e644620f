#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for v in wg2 nolb nohist wg3 wg2; do
  echo "== $v"; LD_PRELOAD=build/ab/$v.so python scripts/bench_sort_quick.py $v --rmat 2>&1 | grep "rmat\|tsamd\]"
done

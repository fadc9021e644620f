#!/bin/bash
# alternating same-box runs: the unfused functional coalesce / transpose (tsamd::sort_coalesce + segment_reduce) first, then the shipped route
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  TSAMD_COALESCE_UNFUSED=1 python scripts/bench_sort_quick.py unfused 2>/dev/null | cut -c1-400
  python scripts/bench_sort_quick.py shipped 2>/dev/null | cut -c1-400
done

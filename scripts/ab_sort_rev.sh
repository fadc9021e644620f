cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  LD_PRELOAD=build/ab/libtsamd_base.so python scripts/bench_sort_quick.py base 2>/dev/null | cut -c1-330
  python scripts/bench_sort_quick.py shipped 2>/dev/null | cut -c1-330
done

#!/bin/bash
# SQ counter pass (issue-side view) over prof_kernels.py workloads: bash scripts/profile_sq.sh TAG workload...
# Output: gpurun_out/TAG/sq/ ; condensed by scripts/sq_table.py.  Counters only (no trace domains) -- one pass.
set -u
TAG=${1:-r04_sq}
shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-include-regex tsamd --output-format csv -d $OUT/sq -o sq -- python scripts/prof_kernels.py "$@" > $OUT/sq.log 2>&1
find $OUT -name "*agent_info*" -delete
F=$(find $OUT/sq -name "*counter_collection.csv" | head -1)
python scripts/sq_table.py $F > $OUT/sq_table.md 2>&1
cat $OUT/sq_table.md

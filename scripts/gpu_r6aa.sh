#!/bin/bash
# round 6 (re-entry): whole GPU suite + smoke on the tree after the sampler / fuzz commits
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6aa
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r6aa/gpu_suite.log

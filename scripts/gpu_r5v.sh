#!/bin/bash
# round 5 (second session): PMC counters per kernel on the current tree -> gpurun_out/r05b_pmc
bash scripts/profile_pmc.sh r05b_pmc > gpurun_out/r05b_pmc.log 2>&1
tail -3 gpurun_out/r05b_pmc.log
du -sh gpurun_out/r05b_pmc

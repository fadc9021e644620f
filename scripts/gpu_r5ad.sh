#!/bin/bash
# SQ counter pass over the round-5 kernels: sort / construct / coalesce, the drop-in min / max training step, C4, stress
bash scripts/profile_sq.sh r05_sq sort_coo_7m5 construct_7m5 coalesce_7m5 c3_matmul_step_val_bf16_F128 c4_spspmm stress_spspmm > gpurun_out/r05_sq.log 2>&1
tail -45 gpurun_out/r05_sq.log | cut -c1-200

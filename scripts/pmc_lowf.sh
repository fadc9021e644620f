#!/bin/bash
# Where does the merge kernel spend its time on narrow feature matrices?  SQ counters for F = 16 and 128.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_lowf
for F in 16 128; do
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo $set | cut -c1-12 | tr ' ' '_')
    rocprofv3 --pmc $set --kernel-include-regex "spmm_merge" --output-format csv -d gpurun_out/pmc_lowf/F${F}_$tag -o p -- python scripts/prof_spmm.py 21 $F sum 3 > gpurun_out/pmc_lowf/F${F}_$tag.log 2>&1
  done
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_lowf/*/p_counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d.split('/')[2], {k: '%.4g' % (sum(v) / len(v)) for k, v in sorted(agg.items())})
PY

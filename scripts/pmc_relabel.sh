#!/bin/bash
# L2 hit/miss and fetch traffic of the merge kernel with and without the relabel step (north star)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_relabel
for m in 0 auto; do
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
    tag=$(echo $set | cut -c1-10 | tr ' ' '_')
    TSAMD_SPMM_RELABEL=$m rocprofv3 --pmc $set --kernel-include-regex "spmm_merge" --output-format csv -d gpurun_out/pmc_relabel/${m}_$tag -o p -- python scripts/prof_spmm.py 21 128 sum 3 > gpurun_out/pmc_relabel/${m}_$tag.log 2>&1
  done
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_relabel/*/p_counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d.split('/')[2], {k: '%.4g' % (sum(v) / len(v)) for k, v in sorted(agg.items())})
PY

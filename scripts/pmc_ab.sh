#!/bin/bash
# A/B PMC comparison of two lib variants on the uniform-degree graph
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_ab
for v in ${VARIANTS:-old base}; do
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
    tag=$(echo $set | cut -c1-12 | tr ' ' '_')
    TSAMD_LIB=build/variants/$v.so rocprofv3 --pmc $set --kernel-include-regex "spmm_(rows|merge|long)" --output-format csv -d gpurun_out/pmc_ab/${v}_$tag -o p -- python scripts/exp_spmm.py $1 > gpurun_out/pmc_ab/${v}_$tag.log 2>&1
  done
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_ab/*/p_counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        k = r['Kernel_Name'].split('(')[0].split('::')[-1][:28]
        agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
    print(d.split('/')[2])
    for (k, c), v in sorted(agg.items()):
        print('   %-28s %-22s n=%d avg=%.4g' % (k, c, len(v), sum(v)/len(v)))
PY

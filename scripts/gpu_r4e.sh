cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04e; cp pytorch_sparse_amd/lib/libtsamd.so build/ab/new.so
echo "default rule" > gpurun_out/r04e/relabel_minmax.log; python scripts/ab_minmax_fw.py new >> gpurun_out/r04e/relabel_minmax.log 2>&1
echo "TSAMD_SPMM_RELABEL=1" >> gpurun_out/r04e/relabel_minmax.log; TSAMD_SPMM_RELABEL=1 python scripts/ab_minmax_fw.py new >> gpurun_out/r04e/relabel_minmax.log 2>&1
echo "TSAMD_SPMM_RELABEL=0" >> gpurun_out/r04e/relabel_minmax.log; TSAMD_SPMM_RELABEL=0 python scripts/ab_minmax_fw.py new >> gpurun_out/r04e/relabel_minmax.log 2>&1
cat gpurun_out/r04e/relabel_minmax.log

// tsamd_spmm_partial: the SpMM kernels of spmm.hip instantiated with the partial-product row sink (combine with what
// the earlier column blocks left in out / arg_out; include/tsamd.h).  A separate translation unit so that the
// instantiations behind every other entry point stay exactly as they were tuned -- and the two compile in parallel.
#define TSAMD_SPMM_PARTIAL_BUILD 1
#include "spmm.hip"

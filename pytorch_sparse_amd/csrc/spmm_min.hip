// The min instantiations of the SpMM kernels of spmm.hip (and their int32-id variants), reached from the main unit through
// tsamd::spmm_min_bridge: a separate translation unit only so that the pieces compile in parallel (spmm.hip alone took
// 176 s -- the longest step of build(); now 39 + 66 + 66 s side by side).  No entry point is defined here.
#define TSAMD_SPMM_TU 2
#include "spmm.hip"

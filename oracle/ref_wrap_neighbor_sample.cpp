// see ref_wrap.h; the include path puts /root/reference/csrc first.
#include "ref_wrap.h"
#include "neighbor_sample.cpp"

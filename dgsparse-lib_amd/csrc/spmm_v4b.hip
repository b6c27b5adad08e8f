// spmm_v4b.hip -- instantiates the V=4 max / min SpMM kernels (arg ids, position-tracked split path).
#define DGS_TU_ARG_ONLY
#include "spmm_impl.h"

namespace dgs {
int spmm_run_v4_arg(int G, const SpmmArgs &a) { return dispatch_g<4>(G, a); }
}  // namespace dgs

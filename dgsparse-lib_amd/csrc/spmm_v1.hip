// spmm_v1.hip -- instantiates the scalar (V=1) SpMM kernels: feature counts that are not a multiple of 4, or
// operands whose base is not 16-byte aligned.
#include "spmm_impl.h"

namespace dgs {
int spmm_run_v1(int G, const SpmmArgs &a) { return dispatch_g<1>(G, a); }
}  // namespace dgs

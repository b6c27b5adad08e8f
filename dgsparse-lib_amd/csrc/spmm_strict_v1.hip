// spmm_strict_v1.hip -- the strict-order sum / mean launches for scalar lanes (V = 1: feature counts that are not a
// multiple of 4, or operands whose base is not 16-byte aligned).
#define DGS_TU_STRICT
#include "spmm_impl.h"

namespace dgs {
int spmm_run_strict_v1(int G, const SpmmArgs &a) { return dispatch_strict<1>(G, a); }
}  // namespace dgs

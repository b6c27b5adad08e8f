// spmm_strict.hip -- instantiates the strict-order sum / mean launches (spmm_strict.h): V = 4 and V = 1, fmaf chain and
// the uncontracted multiply-add chain.
#define DGS_TU_STRICT
#include "spmm_impl.h"

namespace dgs {
int spmm_run_strict(int G, int V, const SpmmArgs &a) { return V == 4 ? dispatch_strict<4>(G, a) : dispatch_strict<1>(G, a); }
}  // namespace dgs

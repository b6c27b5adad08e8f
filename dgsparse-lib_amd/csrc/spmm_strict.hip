// spmm_strict.hip -- instantiates the strict-order sum / mean launches (spmm_strict.h) for 16-byte lanes (V = 4): fmaf chain
// and the uncontracted multiply-add chain.  The scalar-lane instances live in spmm_strict_v1.hip (build parallelism).
#define DGS_TU_STRICT
#include "spmm_impl.h"

namespace dgs {
int spmm_run_strict_v1(int G, const SpmmArgs &a);
int spmm_run_strict(int G, int V, const SpmmArgs &a) { return V == 4 ? dispatch_strict<4>(G, a) : spmm_run_strict_v1(G, a); }
}  // namespace dgs

// tests/emu/rocprim/rocprim.hpp -- TEST INFRASTRUCTURE ONLY: host stand-ins for the three rocPRIM primitives the setup paths use
// (csrc/spmm_plan.hip, csrc/csr2csc.hip), with rocPRIM's calling convention (first call with a null temporary buffer returns the
// size it wants).  See tests/emu/hip/hip_runtime.h.
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>

#include "hip/hip_runtime.h"

namespace rocprim {
template <typename T>
struct plus {
  T operator()(const T &a, const T &b) const { return a + b; }
};
template <typename In, typename Out, typename Init, typename Op>
inline hipError_t exclusive_scan(void *tmp, size_t &bytes, In in, Out out, Init init, size_t n, Op op, hipStream_t = nullptr,
                                 bool = false) {
  if (!tmp) {
    bytes = 256;
    return hipSuccess;
  }
  auto acc = init;
  for (size_t i = 0; i < n; i++) {
    const auto v = in[i];  // in and out may alias
    out[i] = acc;
    acc = op(acc, v);
  }
  return hipSuccess;
}
template <typename KI, typename K, typename VI, typename V>
inline hipError_t radix_sort_pairs(void *tmp, size_t &bytes, KI *kin, K *kout, VI *vin, V *vout, size_t n, unsigned begin_bit = 0,
                                   unsigned end_bit = 8 * sizeof(K), hipStream_t = nullptr, bool = false) {
  if (!tmp) {
    bytes = 256;
    return hipSuccess;
  }
  typedef typename std::make_unsigned<K>::type UK;
  const UK mask = end_bit >= 8 * sizeof(K) ? ~UK(0) : ((UK(1) << end_bit) - 1);
  std::vector<size_t> idx(n);
  std::iota(idx.begin(), idx.end(), size_t(0));
  std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return (((UK)kin[a] & mask) >> begin_bit) < (((UK)kin[b] & mask) >> begin_bit); });
  std::vector<K> k2(n);
  std::vector<V> v2(n);
  for (size_t i = 0; i < n; i++) {
    k2[i] = kin[idx[i]];
    v2[i] = vin[idx[i]];
  }
  std::copy(k2.begin(), k2.end(), kout);
  std::copy(v2.begin(), v2.end(), vout);
  return hipSuccess;
}
}  // namespace rocprim

// Host-side check of the work deal the strict / hub blocks use (csrc/spmm_strict.h: strict_deal): every (group, slice) task of
// every segment is taken by exactly one worker, groups stay on the XCD g % nx, and the rotation carried across segments
// keeps the deal a permutation.  Built and run by tests/test_host_cpu.py (no GPU needed: nothing here touches a device).
#define DGS_TU_STRICT
#include "spmm_impl.h"

#include <cstdio>
#include <map>
#include <vector>

int main() {
  int bad = 0;
  const int nbs[] = {8, 16, 24, 256, 264, 2048, 12, 1};
  const int segs[][2] = {{0, 4}, {1, 4}, {5, 4}, {105, 4}, {1000, 2}, {37, 1}, {8, 4}, {413, 4}};
  for (int nblocks : nbs) {
    const int nx = (nblocks & 7) == 0 ? 8 : 1;
    for (int workers_per_block : {1, 4}) {  // cooperative hub tasks: one worker per block; wave-level tasks: four
      std::vector<std::map<long long, int>> seen(8);
      const int nslots = (nblocks / nx) * workers_per_block;
      for (int bid = 0; bid < nblocks; bid++)
        for (int w = 0; w < workers_per_block; w++) {
          const int x = bid % nx, slot = (bid / nx) * workers_per_block + w;
          int rot = 0;
          for (int sgi = 0; sgi < 8; sgi++)
            dgs::strict_deal(segs[sgi][0], segs[sgi][1], x, nx, slot, nslots, rot, [&](int g, int j) {
              if (g < 0 || g >= segs[sgi][0] || j < 0 || j >= segs[sgi][1] || g % nx != x) bad++;
              seen[sgi][(long long)g * 16 + j]++;
            });
        }
      for (int sgi = 0; sgi < 8; sgi++) {
        if ((int)seen[sgi].size() != segs[sgi][0] * segs[sgi][1]) bad++;
        for (auto &kv : seen[sgi])
          if (kv.second != 1) bad++;
      }
    }
  }
  // hub tables: class regions do not overlap and hold every row a class can have
  for (long long nnz : {1000LL, 1LL << 20, 16108469LL, 2147483000LL})
    for (int thub : {1024, 2048, 8192, 65534}) {
      const dgs::HubTab t = dgs::hub_tab(nnz, thub, nnz / 64 + 2);
      for (int c = 0; c < dgs::kHubClasses; c++) {
        const long long cap = (c + 1 < dgs::kHubClasses ? t.base[c + 1] : t.base[0] + dgs::hub_tab_entries(nnz, thub)) - t.base[c];
        if (cap < nnz / ((long long)thub << c) + 1) bad++;
      }
      for (int len : {thub + 1, 2 * thub, 2 * thub + 1, 32 * thub, 33 * thub, 2000000000})
        if (len > thub) {
          const int c = dgs::hub_class(len, thub);
          if (c < 0 || c >= dgs::kHubClasses || len <= (thub << c) / 1 * 1 - 0 * 0 && c > 0) bad += (len <= ((long long)thub << c));
          if (c + 1 < dgs::kHubClasses && len > ((long long)thub << (c + 1))) bad++;
        }
    }
  printf("%s\n", bad ? "FAILED" : "ok");
  return bad ? 1 : 0;
}

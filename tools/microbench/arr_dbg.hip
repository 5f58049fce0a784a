// prepack one layer with the greedy deal alone (packed_arrange = 2) and with the local search on top (1) and count the words that
// differ: 0 means pk_improve_kernel took no swap (what a miscompiled cost difference once did).  make -C tools/microbench arr_dbg
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../include/aqlm_hip.h"
int main() {
  const int in = 4096, out = 4096;
  std::vector<uint16_t> codes((size_t)out * in / 8);
  uint32_t r = 12345;
  for (auto& c : codes) { r = r * 1664525u + 1013904223u; c = (uint16_t)(r >> 16); }
  void *dc, *p1, *p2;
  hipMalloc(&dc, codes.size() * 2);
  hipMemcpy(dc, codes.data(), codes.size() * 2, hipMemcpyHostToDevice);
  const size_t pb = aqlm_hip_prepack_1x16_bytes(out, in, 8);
  hipMalloc(&p1, pb); hipMalloc(&p2, pb);
  hipMemset(p1, 0, pb); hipMemset(p2, 0, pb);
  aqlm_hip_packed_desc d1, d2;
  aqlm_hip_set_tuning("packed_arrange", 2);
  int rc1 = aqlm_hip_prepack_1x16(dc, out, in, 8, p1, pb, &d1, nullptr);
  aqlm_hip_set_tuning("packed_arrange", 1);
  int rc2 = aqlm_hip_prepack_1x16(dc, out, in, 8, p2, pb, &d2, nullptr);
  hipDeviceSynchronize();
  std::vector<uint32_t> h1(pb / 4), h2(pb / 4);
  hipMemcpy(h1.data(), p1, pb, hipMemcpyDeviceToHost);
  hipMemcpy(h2.data(), p2, pb, hipMemcpyDeviceToHost);
  size_t diff = 0;
  for (size_t i = 0; i < h1.size(); ++i) diff += h1[i] != h2[i];
  printf("rc %d %d used %u %u differing words %zu of %zu\n", rc1, rc2, d1.used_bytes, d2.used_bytes, diff, h1.size());
  return 0;
}

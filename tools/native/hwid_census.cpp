// Census of the HW_ID register fields under a conv-like launch (256-thread workgroups, 4 per CU): which field tells the
// co-resident workgroups of a CU apart?  Build: hipcc --offload-arch=gfx950 -O2 tools/native/hwid_census.cpp -o tools/native/hwid_census.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 4) void census(unsigned* out, unsigned long long* t) {
  __shared__ float pad[4096];
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID, 32 bits
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = id;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    if (threadIdx.x == 0) t[blockIdx.x] = __builtin_readcyclecounter();
  }
  // stay resident long enough that the whole first round overlaps
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(100);
  if (pad[(threadIdx.x + 1) & 255] < 0) out[0] = 0;
}
int main() {
  const int nb = 1024;
  unsigned* d; unsigned long long* t;
  hipMalloc(&d, nb * 8 * sizeof(unsigned)); hipMalloc(&t, nb * 8);
  hipLaunchKernelGGL(census, dim3(nb), dim3(256), 0, 0, d, t);
  std::vector<unsigned> h(nb * 8); std::vector<unsigned long long> ht(nb);
  hipMemcpy(h.data(), d, nb * 8 * sizeof(unsigned), hipMemcpyDeviceToHost);
  hipMemcpy(ht.data(), t, nb * 8, hipMemcpyDeviceToHost);
  // print the first 12 blocks raw, then the distribution of low fields
  for (int b = 0; b < 12; ++b) {
    printf("block %4d xcc %u:", b, h[b * 8 + 1]);
    for (int w = 0; w < 4; ++w) { unsigned v = h[(b * 4 + w) * 2]; printf("  w%d id=%08x wave=%u simd=%u pipe=%u cu=%u sh=%u se=%u", w, v, v & 15, (v >> 4) & 3, (v >> 6) & 3, (v >> 8) & 15, (v >> 12) & 1, (v >> 13) & 7); }
    printf("\n");
  }
  std::map<unsigned, std::map<unsigned, int>> percu;   // (xcc, se, sh, cu) -> wave_id of wave 0 -> count
  for (int b = 0; b < nb; ++b) { unsigned v = h[b * 8], x = h[b * 8 + 1]; percu[(x << 16) | (v & 0xff00)][v & 15]++; }
  printf("distinct CUs seen: %zu\n", percu.size());
  int shown = 0;
  for (auto& kv : percu) { if (shown++ >= 6) break; printf("cu key %06x:", kv.first); for (auto& w : kv.second) printf(" wave_id %u x%d", w.first, w.second); printf("\n"); }
  return 0;
}

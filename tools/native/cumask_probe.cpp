// Which CUs does a stream created with hipExtStreamCreateWithCUMask use on MI355X (8 XCDs x 32 CUs)?  Launches a census kernel on
// streams with different masks and prints, per XCC_ID, how many distinct CUs ran workgroups.
// Build: hipcc --offload-arch=gfx950 -O2 tools/native/cumask_probe.cpp -o tools/native/cumask_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <set>
#include <vector>
__global__ __launch_bounds__(256) void census(unsigned* out) {
  const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = id; out[blockIdx.x * 2 + 1] = xcc; }
  for (int i = 0; i < 300; ++i) __builtin_amdgcn_s_sleep(100);
}
static void run(const char* name, hipStream_t st) {
  const int nb = 4096;
  unsigned* d;
  hipMalloc(&d, nb * 2 * sizeof(unsigned));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, st);
  hipLaunchKernelGGL(census, dim3(nb), dim3(256), 0, st, d);
  hipEventRecord(b, st);
  hipStreamSynchronize(st);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned> h(nb * 2);
  hipMemcpy(h.data(), d, nb * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> cus;
  for (int i = 0; i < nb; ++i) cus[h[i * 2 + 1] & 15].insert(h[i * 2] & 0xff00);      // (se, sh, cu) fields
  printf("%-28s %7.2f ms  XCDs used %zu:", name, ms, cus.size());
  size_t total = 0;
  for (auto& kv : cus) { printf(" x%u:%zu", kv.first, kv.second.size()); total += kv.second.size(); }
  printf("  total CUs %zu\n", total);
  hipFree(d);
}
int main() {
  run("default stream", 0);
  struct { const char* name; uint32_t m[8]; } cases[] = {
    {"low 128 bits", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0}},
    {"high 128 bits", {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}},
    {"even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
    {"low 16 of every 32", {0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu}},
    {"first word only", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
  };
  for (auto& c : cases) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, c.m);
    if (e != hipSuccess) { printf("%s: create failed: %s\n", c.name, hipGetErrorString(e)); continue; }
    run(c.name, st);
    hipStreamDestroy(st);
  }
  return 0;
}

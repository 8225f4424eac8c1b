// Sustained matrix-pipe rate under the board's power management: every SIMD issues back-to-back independent MFMAs (no memory traffic);
// reports TFLOP/s from event time and the mean shader cycles per MFMA per wave (s_memtime is a constant 100 MHz counter, clock64 the shader clock).
//   hipcc --offload-arch=gfx950 -O2 mfma_peak.cpp -o mfma_peak.bin ; ./mfma_peak.bin [seconds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int KIND>   // 0: v_mfma_f32_32x32x2_f32   1: v_mfma_f32_32x32x16_bf16
__global__ __launch_bounds__(256) void peak_kernel(float* out, int iters, unsigned long long* cyc) {
  f32x16_t acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  bf16x8_t ab, bb;
  for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)a; bb[j] = (__bf16)b; }
  const unsigned long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
      }
  }
  const unsigned long long c1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 123.456f) out[threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 1.0;
  float* out; unsigned long long* cyc; CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int kind = 0; kind < 2; ++kind) {
    const int iters = 20000, grid = 256 * 2;          // 2 workgroups x 4 waves per CU = 2 waves per SIMD
    const double flops_per_mfma = kind == 0 ? 32.0 * 32 * 2 * 2 : 32.0 * 32 * 16 * 2;
    const double flops = (double)grid * 4 * iters * 16 * flops_per_mfma;
    double total_ms = 0; int n = 0; float last = 0;
    while (total_ms < secs * 1e3) {
      CK(hipEventRecord(e0));
      if (kind == 0) peak_kernel<0><<<grid, 256>>>(out, iters, cyc); else peak_kernel<1><<<grid, 256>>>(out, iters, cyc);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); total_ms += ms; last = ms; ++n;
      if (n == 1 || n == 2) printf("  launch %d: %.2f ms -> %.1f TFLOP/s\n", n, ms, flops / ms / 1e9);
    }
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%s: %d launches, last %.2f ms -> %.1f TFLOP/s; wave 0: %.1f shader cycles per MFMA per SIMD (two waves share it), implied clock %.0f MHz\n",
           kind == 0 ? "v_mfma_f32_32x32x2_f32 " : "v_mfma_f32_32x32x16_bf16", n, last, flops / last / 1e9, (double)c / (iters * 16) / 2, (double)c / (last * 1e-3) / 1e6);
  }
  return 0;
}

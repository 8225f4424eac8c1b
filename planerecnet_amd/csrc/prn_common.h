// Shared helpers for the gfx950 kernels (wave64, 256 CUs in 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/prn.h"

#define PRN_WAVE 64

void prn_set_error(const char* fmt, ...);

#define PRN_CHECK_LAUNCH(name)                                                        \
  do {                                                                                \
    hipError_t e_ = hipGetLastError();                                                \
    if (e_ != hipSuccess) {                                                           \
      prn_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));            \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

#define PRN_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      prn_set_error(__VA_ARGS__);       \
      return 2;                         \
    }                                   \
  } while (0)

// Block b is observed to run on XCD b % 8 (MI355X_MICROARCH.md): give every XCD a contiguous range of
// logical tile ids so that neighbouring tiles (which share operand rows) hit the same 4 MiB L2.
// Speed only -- any placement is correct.
__device__ __forceinline__ int prn_xcd_remap(int bid, int nblocks) {
  const int per = nblocks >> 3;
  if (per == 0 || bid >= (per << 3)) return bid;
  return (bid & 7) * per + (bid >> 3);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Internal (not part of the C ABI): the fixed-order split reductions of prn_conv.hip, shared with prn_dcnv2.hip.
//   y = epi(sum_s ws[s] + bias[m] + addend)  over total = B*M*HoWo elements  /  out[i] = sum_s ws[s][i] over n elements
int prn_launch_reduce_epilogue(const float* ws, const float* bias, const float* addend, float* y, int64_t total, int M, int HoWo, int splits,
                               int epi, hipStream_t st);
int prn_launch_reduce_splits(const float* ws, float* out, int64_t n, int splits, hipStream_t st);
int prn_quantise_splits(int64_t tiles, int splits);
//   Y_z[M x P] = epi(U_z[M x C] * V_z[C x P]) for z < nb in one launch (prn_gemm_batched with an activation)
int prn_gemm_batched_epi(int M, int C, int P, int nb, const float* U, const float* V, float* Y, int epi, void* stream);

// fp32 GEMM on the bf16 matrix pipe by exact three-way operand splitting (prn_gemm_split.hip):
//   y[z][b][m][p] = epi(sum_k w[z][m][k] * x[z][b][k][p] + bias[m] + addend)   for z < nz, b < B  (a (z, b) image of x is [K][HW], of y [M][HW])
// prn_split_gemm_plan: 0 = keep the fp32 MFMA kernel, else the number of K splits to run it with.
int prn_split_gemm_plan(int M, int K, int B, int HW, int nz);
int64_t prn_split_gemm_image_bytes(int M, int K, int nz);
int64_t prn_split_gemm_partial_bytes(int M, int B, int HW, int nz, int splits);
int prn_split_gemm(const float* w, const float* x, const float* bias, const float* addend, float* y, void* images, float* partial, int M, int K, int B, int HW,
                   int nz, int64_t zw, int64_t zx, int64_t zy, int epi, int splits, hipStream_t st, int phase);
void* prn_split_scratch(hipStream_t st, int64_t bytes);
// DCNv2 forward on the fp16-piece split kernel (prn_gemm_split.hip): plan = K splits (0: keep the fp32 kernel), workspace, launch
int prn_split_dcn_plan(int M, int K, int N);
int64_t prn_split_dcn_ws_bytes(int M, int K, int B, int HoWo, int splits);
int prn_split_dcn_fwd(const float* w, const float* x, const void* table, const float* bias, float* y, void* ws, int B, int C, int HW, int M, int HoWo, int nchunks,
                      int epi, int splits, hipStream_t st, int phase);

// Shared helpers for the gfx950 kernels (wave64, 256 CUs in 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/prn.h"

#define PRN_WAVE 64

void prn_set_error(const char* fmt, ...);

// Tuning overrides from the environment (DESIGN.md lists them): read ONCE per switch, through a function-local `static const` -- C++11 makes that
// initialisation thread-safe (the backward entry points are called from autograd's threads), and nothing ever writes them again.
#include <stdlib.h>
static inline int prn_env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
struct prn_env4 { int v[4]; };
static inline prn_env4 prn_env_ints(const char* name) {      // "a,b,c,d" (missing fields: 0)
  prn_env4 r = {{0, 0, 0, 0}};
  if (const char* e = getenv(name)) sscanf(e, "%d,%d,%d,%d", &r.v[0], &r.v[1], &r.v[2], &r.v[3]);
  return r;
}

// Measurement aid (include/prn.h: prn_debug_skip_launches): launches the caller asked to leave out.  0 in every product run.
extern int prn_skip_mask;
#define PRN_SKIPPED(bit) (__builtin_expect(prn_skip_mask & (bit), 0))
#define PRN_REPS(bit) (__builtin_expect(prn_skip_mask & ((bit) | ((bit) << 8)), 0) ? ((prn_skip_mask & (bit)) ? 0 : 2) : 1)   // launches of a family: 1; left out: 0; issued twice: 2

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

// i -> (x = i % W, y = (i / W) % H, z = i / (W * H)) for the one-element-per-thread kernels.  With i as int64_t the three 64-bit divisions were
// most of such a kernel's instructions (the generic resize adjoint: ~350 VALU instructions per pixel); every tensor of this network indexes in
// 32 bits, where a division is a float reciprocal and a correction.
__device__ __forceinline__ void prn_idx3(int64_t i, int W, int H, int& x, int& y, int64_t& z) {
  if (i < (1LL << 31)) {
    const unsigned u = (unsigned)i, q = u / (unsigned)W, q2 = q / (unsigned)H;
    x = (int)(u - q * (unsigned)W); y = (int)(q - q2 * (unsigned)H); z = q2;
  } else {
    x = (int)(i % W); y = (int)((i / W) % H); z = i / ((int64_t)W * H);
  }
}

// PRN_EPT elements per thread, a whole grid apart (every load instruction stays as coalesced as with one), all loads issued before the first store.
// Pays where an element needs MANY loads (the x2 resize adjoint: sixteen, 72 -> 40 us); measured useless for the one-load-per-element folds, which are
// bound by bytes per memory instruction, not by loads in flight (they got the four-pixels-per-thread form instead).
// four consecutive floats at a 4-byte aligned address: ONE global_load_dwordx4 (gfx950 global memory takes unaligned vector accesses)
struct __attribute__((packed, aligned(4))) prn_f4u { float v[4]; };
constexpr int PRN_EPT = 4;
__host__ __device__ inline unsigned prn_ept_blocks(int64_t n) { return (unsigned)((((n + PRN_EPT - 1) / PRN_EPT) + 255) / 256); }
#define PRN_EPT_BEGIN(total_) \
  const int64_t T_ = (int64_t)gridDim.x * 256, i0_ = (int64_t)blockIdx.x * 256 + threadIdx.x; \
  float v_[PRN_EPT]; \
  _Pragma("unroll") for (int k_ = 0; k_ < PRN_EPT; ++k_) { \
    const int64_t i = (i0_ + k_ * T_ < (total_)) ? i0_ + k_ * T_ : (total_) - 1;          /* past the end: recompute the last element, never stored */
#define PRN_EPT_END(total_, out_) \
  } \
  _Pragma("unroll") for (int k_ = 0; k_ < PRN_EPT; ++k_) if (i0_ + k_ * T_ < (total_)) (out_)[i0_ + k_ * T_] = v_[k_];


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

// (a, b) summed over the workgroup in a fixed order (wave shuffles, then the waves' totals in wave order); sm: 2 * (waves per block) doubles
__device__ __forceinline__ void block_sum2(double& a, double& b, double* sm) {
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sm[w] = a; sm[nw + w] = b; }
  __syncthreads();
  double ra = 0.0, rb = 0.0;
  for (int i = 0; i < nw; ++i) { ra += sm[i]; rb += sm[nw + i]; }
  a = ra; b = rb;
}

// Winograd F(4x4, 3x3), one pass over a row or column of a 6-vector (prn_winograd.hip): o = B^T v (input side), o = A^T m (output side)
__device__ __forceinline__ void bt6(const float* v, float* o) {
  o[0] = 4.f * v[0] - 5.f * v[2] + v[4];
  o[1] = -4.f * (v[1] + v[2]) + v[3] + v[4];
  o[2] = 4.f * (v[1] - v[2]) - v[3] + v[4];
  o[3] = -2.f * v[1] - v[2] + 2.f * v[3] + v[4];
  o[4] = 2.f * v[1] - v[2] - 2.f * v[3] + v[4];
  o[5] = 4.f * v[1] - 5.f * v[3] + v[5];
}
__device__ __forceinline__ void at6(const float* m, float* o) {
  const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
  o[0] = m[0] + s12 + s34;
  o[1] = d12 + 2.f * d34;
  o[2] = s12 + 4.f * s34;
  o[3] = d12 + 8.f * d34 + m[5];
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Internal (not part of the C ABI): the fixed-order split reductions of prn_conv.hip, shared with prn_dcnv2.hip.
//   y = epi(sum_s ws[s] + bias[m] + addend)  over total = B*M*HoWo elements  /  out[i] = sum_s ws[s][i] over n elements
int prn_launch_reduce_epilogue(const float* ws, const float* bias, const float* addend, float* y, int64_t total, int M, int HoWo, int splits,
                               int epi, hipStream_t st);
int prn_launch_reduce_splits(const float* ws, float* out, int64_t n, int splits, hipStream_t st);
int prn_quantise_splits(int64_t tiles, int splits);
//   Y_z[M x P] = epi(U_z[M x C] * V_z[C x P]) for z < nb in one launch (prn_gemm_batched with an activation)
int prn_gemm_batched_epi(int M, int C, int P, int nb, const float* U, const void* u_images, const float* V, float* Y, void* ws, const prn_gemm_opts* opts, int epi,
                         void* stream);

// fp32 GEMM on the 16-bit matrix pipe by operand splitting (prn_gemm_split.hip):
//   y[z][b][m][p] = epi(sum_k w[z][m][k] * x[z][b][k][p] + bias[m] + addend)   for z < nz, b < B  (a (z, b) image of x is [K][HW], of y [M][HW])
// prn_split_gemm_plan: 0 = keep the fp32 MFMA kernel, else the number of K splits to run it with (opts == NULL: 0).
int prn_split_gemm_plan(int M, int K, int B, int HW, int nz, const prn_gemm_opts* opts);
int64_t prn_split_gemm_image_bytes(int M, int K, int nz);
int64_t prn_split_gemm_partial_bytes(int M, int B, int HW, int nz, int splits);
// w_images: the caller's current images of w, or NULL: w is cut into `images_ws` (prn_split_gemm_image_bytes) by this call.
int prn_split_gemm(const float* w, const void* w_images, const float* x, const float* bias, const float* addend, float* y, void* images_ws, float* partial, int M,
                   int K, int B, int HW, int nz, int64_t zw, int64_t zx, int64_t zy, int epi, int splits, const prn_gemm_opts* opts, hipStream_t st, int phase);
// the same kernel with a zero-padded KH x KW gather as activation operand (tap-major weight images cut per call; fp16 pieces, C % 32 == 0)
int prn_split_conv_taps(const float* w, const float* x, const float* bias, const float* addend, float* y, void* images_ws, float* partial, int M, int C, int B, int XH,
                        int XW, int Ho, int Wo, int KH, int KW, int stride, int pad, int epi, int splits, const prn_gemm_opts* o, hipStream_t st, int phase);
// ... and with the four sub-pixel phases of PRN_IN_UP2_PHASE as its z axis (wp [4][M][C][2][2] -> y [B][M][2H][2W]; no K split)
int prn_split_conv_up2(const float* wp, const float* x, const float* bias, float* y, void* images_ws, int M, int C, int B, int H, int W, int epi, const prn_gemm_opts* o,
                       hipStream_t st);
// weight-gradient plan knobs of a call (NULL opts: zeros)
static inline prn_gemm_opts prn_opts_or_zero(const prn_gemm_opts* o) {
  prn_gemm_opts z;
  if (o) return *o;
  z.split_mode = 0; z.split_kind = 0; z.split_products = 0; z.split_min_tiles = 0; z.split_min_gflop = 0.f; z.wgrad_wgs = 0; z.wgrad_target = 0; z.wgrad_split = 0;
  return z;
}

// Weight gradient of a plain GEMM layer on the 16-bit pipe (prn_wgrad16.hip): dW[z][m][c] = sum_n dy[z][m][n] x[z][c][n], n = (image, pixel).
// plan: pixel splits (0: keep the fp32 kernel); launch: out = dw (splits == 1) or partials [splits][nz][M][C].
int prn_wgrad16_plan(int M, int C, int64_t N, int HW, int nz, const prn_gemm_opts* opts);
int prn_wgrad16_launch(const float* dy, const float* x, const float* const* gdy, const float* const* gx, int ngroup, float* out, int M, int C, int B, int HW, int nz,
                       int64_t zdy, int64_t zx, int splits, const prn_gemm_opts* opts, hipStream_t st);

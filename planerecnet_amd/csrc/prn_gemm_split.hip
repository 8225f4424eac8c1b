// fp32 GEMM on the bf16 matrix pipe by EXACT three-way operand splitting -- the 1x1 convolutions (reference models/backbone.py:56-66,
// models/fpn.py:46-57) and the batched transform-domain products of the Winograd path, i.e. the plain-GEMM launches of conv_igemm_kernel.
//
//   x = h + m + l,  h / m / l = the three consecutive 8-bit slices of x's 24-bit significand (truncation, not rounding): every piece is
//   exactly a bf16 and the three add up to x exactly.  a*b is then the sum of nine piece products, each EXACT in fp32 (8 x 8 bits);
//   the kernel issues six of them as v_mfma_f32_32x32x16_bf16 (fp32 accumulate) and drops m*l, l*m, l*l, which are <= 2^-23 |a*b| --
//   below one rounding of the fp32 accumulation.  Measured against fp64 on the step's shapes the result is as close as the fp32 MFMA's
//   (max 2.2e-7 / rms 1.8e-8 of sum|a||b| against 2.7e-7 / 2.2e-8; tools/native/gemm_split_lab, tests/test_ops_gpu.py), at 6 x 1/16 = 3/8 of
//   the fp32 pipe's issue time per MAC.  NOT identical in every sense: the 16-bit pipe's accumulate step truncates toward -infinity, a coherent
//   -1e-9 of sum|a||b| per output that round 4's model-level runs exposed; see the sign pattern in split16_gemm_kernel.
//
// Layout.  y[z][b][m][p] = sum_k w[z][m][k] * x[z][b][k][p]; one workgroup = 128 rows x 128 pixels of one (z, b) image, four waves of
// 128 rows x 32 pixels.  The weight side is split ONCE per launch by split_prepare_kernel into "images" [m tile][k slice][piece][k group]
// [row][8 x bf16] that are exactly the LDS image a workgroup needs per 32-deep K slice (24 KB), so the A stream is buffer_load ... lds
// (no registers, no ds_write) into a two-stage ring.  The activation side never touches LDS: a lane of the 32x32x16 MFMA holds eight
// consecutive k of ONE pixel, and with NCHW activations lanes = pixels is the coalesced direction, so every wave fetches its own B
// operand with 16 dword buffer loads per slice (rows beyond K read 0: the row is part of the VGPR offset, which the descriptor's range check sees), splits it in registers (5.5 VALU
// per element, under the MFMAs of the slice before) and feeds the pieces straight to the matrix pipe.  One barrier per slice.
#include "prn_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BK = 32;
constexpr int IMG_U4 = 1536;      // uint4 per image: 3 pieces x 4 k-groups x 128 rows x 16 B = 24 KB

// three exact bf16 pieces of two consecutive-k values, packed (low half = first value)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
  const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
  const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
  const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
  h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
  m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
  l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

// w[z][m][k] (row stride K, z stride zw) -> images [z][m tile][k slice][piece][k group][row][8 x bf16], zero padded in M and K
__global__ void split_prepare_kernel(const float* __restrict__ w, uint4* __restrict__ img, int M, int K, long long zw, int mtiles, int kslices, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // one thread per (z, m tile, k slice, k group, row)
  if (i >= total) return;
  const int g = i % 4; const int r = (i / 4) % 128; const long long t = i / 512;      // k group fastest: four lanes read 128 contiguous bytes of one weight row
  const int ks = t % kslices; const long long zm = t / kslices; const int mt = zm % mtiles; const long long z = zm / mtiles;
  const int m = mt * 128 + r, k0 = ks * 32 + g * 8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (m < M && k0 + j < K) ? w[z * zw + (long long)m * K + k0 + j] : 0.f;
  uint4 h, mm, l;
  split2(v[0], v[1], h.x, mm.x, l.x); split2(v[2], v[3], h.y, mm.y, l.y); split2(v[4], v[5], h.z, mm.z, l.z); split2(v[6], v[7], h.w, mm.w, l.w);
  uint4* o = img + t * IMG_U4;
  o[(0 * 4 + g) * 128 + r] = h; o[(1 * 4 + g) * 128 + r] = mm; o[(2 * 4 + g) * 128 + r] = l;
}

__device__ __forceinline__ i32x4_t make_desc(const void* p, unsigned bytes) {
  const unsigned long long q = (unsigned long long)p;
  i32x4_t d;
  d.x = __builtin_amdgcn_readfirstlane((int)(unsigned)q);
  d.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(q >> 32) & 0xffff);
  d.z = __builtin_amdgcn_readfirstlane((int)bytes);
  d.w = 0x00020000;
  return d;
}
// LDS-DMA: 64 lanes x 16 bytes from (descriptor, lane offset + scalar offset) to LDS [lds_byte_addr + 16 * lane].  An asm statement on
// purpose: hipcc neither has to model M0 nor counts it in its own vmcnt bookkeeping; it is ordered by the buffer loads issued after it
// (vmcnt retires in order) and the workgroup barrier.
__device__ __forceinline__ void lds_dma16(unsigned lds_byte_addr, i32x4_t desc, unsigned voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane((int)lds_byte_addr)), "v"(voff), "s"(desc),
               "s"(__builtin_amdgcn_readfirstlane(soff))
               : "memory");
}

struct SplitArgs {
  const uint4* img; const float* x; const float* bias; const float* addend; float* y; float* partial;
  int M, K, B, HW, epi, mtiles, kslices, ptiles, total, splits;
  long long zx, zy;               // element strides of x / y per z (a (z, b) image is K*HW / M*HW elements)
  long long slice;                // elements of one partial slice (splits > 1): nz * B * M * HW
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void split_gemm_kernel(const SplitArgs a) {
  __shared__ uint4 lds[2 * IMG_U4];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int id = prn_xcd_remap(blockIdx.x, a.total);            // the m tiles of one pixel tile share an XCD's L2
  const int mt = id % a.mtiles; const int rest = id / a.mtiles;
  const int pt = rest % a.ptiles; const int zb = rest / a.ptiles; const int b = zb % a.B, z = zb / a.B;
  const int sp = blockIdx.y;
  const int ks0 = (int)((long long)a.kslices * sp / a.splits), ks1 = (int)((long long)a.kslices * (sp + 1) / a.splits);
  const int HW = a.HW, M = a.M;
  const int r = lane & 31, gs = lane >> 5;
  const int px = pt * 128 + wave * 32 + r;
  const int pxc = px < HW ? px : HW - 1;
  const float* xb = a.x + (long long)z * a.zx + (long long)b * a.K * HW;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, a.K * HW * 4, 0x00020000);
  const int xoff = (pxc + gs * 8 * HW) * 4;                      // this lane's byte offset inside a 16-row group
  const uint4* ag = a.img + ((long long)(z * a.mtiles + mt) * a.kslices) * IMG_U4;
  const i32x4_t adesc = make_desc(ag, (unsigned)a.kslices * IMG_U4 * 16u);
  const unsigned lds0 = (unsigned)(unsigned long long)(void*)lds;
  float rn[16];
  bf16x8_t bp[2][3];
  const unsigned podd = (unsigned)__builtin_popcount((unsigned)px) & 1u;                            // see split16_gemm_kernel: Thue-Morse pixels enter negated
  const unsigned sflip16 = podd ? 0x80008000u : 0u, sflip32 = podd ? 0x80000000u : 0u;
#define SPLIT_DMA(ks_, st_) do { \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) \
      lds_dma16(lds0 + (unsigned)(st_) * (IMG_U4 * 16) + (unsigned)(i * 4 + wave) * 1024u, adesc, (unsigned)lane * 16u, ((ks_) * IMG_U4 + (i * 4 + wave) * 64) * 16); \
  } while (0)
#define SPLIT_LOADB(ks_) do { \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) \
        rn[s2 * 8 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xoff + ((ks_) * BK + s2 * 16 + j) * HW * 4, 0, 0)); \
  } while (0)
#define SPLIT_PIECES() do { \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) { \
      uint4 h, m, l; \
      split2(rn[s2 * 8 + 0], rn[s2 * 8 + 1], h.x, m.x, l.x); split2(rn[s2 * 8 + 2], rn[s2 * 8 + 3], h.y, m.y, l.y); \
      split2(rn[s2 * 8 + 4], rn[s2 * 8 + 5], h.z, m.z, l.z); split2(rn[s2 * 8 + 6], rn[s2 * 8 + 7], h.w, m.w, l.w); \
      h.x ^= sflip16; h.y ^= sflip16; h.z ^= sflip16; h.w ^= sflip16; m.x ^= sflip16; m.y ^= sflip16; m.z ^= sflip16; m.w ^= sflip16; \
      l.x ^= sflip16; l.y ^= sflip16; l.z ^= sflip16; l.w ^= sflip16; \
      bp[s2][0] = __builtin_bit_cast(bf16x8_t, h); bp[s2][1] = __builtin_bit_cast(bf16x8_t, m); bp[s2][2] = __builtin_bit_cast(bf16x8_t, l); \
    } } while (0)
  // products smallest first: l*h, h*l, m*m, m*h, h*m, h*h
#define SPLIT_STEP(s2_) do { \
    const int kg = 2 * (s2_) + gs; \
    const bf16x8_t bh = bp[s2_][0], bm = bp[s2_][1], bl = bp[s2_][2]; \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { \
      const uint4* ap = lds + st * IMG_U4 + kg * 128 + i * 32 + r; \
      const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, ap[0]), am = __builtin_bit_cast(bf16x8_t, ap[512]), al = __builtin_bit_cast(bf16x8_t, ap[1024]); \
      f32x16_t c = acc[i]; \
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0); \
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0); \
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0); \
      acc[i] = c; \
    } } while (0)
  f32x16_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  SPLIT_DMA(ks0, 0);
  SPLIT_LOADB(ks0);
  SPLIT_PIECES();
  for (int ks = ks0; ks < ks1; ++ks) {
    const int st = (ks - ks0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (already true: the pieces above waited for loads younger than the DMA)
    __syncthreads();
    const bool more = ks + 1 < ks1;
    const int kn = more ? ks + 1 : ks;
    if (more) SPLIT_DMA(ks + 1, st ^ 1);
    SPLIT_LOADB(kn);
    __builtin_amdgcn_sched_barrier(0);                           // keep the loads up here: hipcc otherwise sinks them next to their first use
    SPLIT_STEP(0);
    __builtin_amdgcn_sched_barrier(0);
    SPLIT_STEP(1);
    SPLIT_PIECES();                                              // next slice's pieces, interleaved with the second step's MFMAs
  }
#undef SPLIT_DMA
#undef SPLIT_LOADB
#undef SPLIT_PIECES
#undef SPLIT_STEP
  const bool cok = px < HW;
  if (a.splits > 1) {                                            // K split: raw partial sums, summed in fixed order by reduce_epilogue_kernel
    float* pb = a.partial + (long long)sp * a.slice + ((long long)z * a.B + b) * (long long)M * HW;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = mt * BM + i * 32 + gs * 4 + (e >> 2) * 8 + (e & 3);
        if (cok && row < M) pb[(long long)row * HW + px] = __uint_as_float(__float_as_uint(acc[i][e]) ^ sflip32);
      }
    return;
  }
  float* yb = a.y + (long long)z * a.zy + (long long)b * M * HW;
  const float* ab = a.addend ? a.addend + (long long)z * a.zy + (long long)b * M * HW : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rbase = mt * BM + i * 32 + gs * 4;
    float bv[16], av[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = rbase + (e >> 2) * 8 + (e & 3); const int rc = row < M ? row : M - 1;
      bv[e] = a.bias ? a.bias[rc] : 0.f; av[e] = ab ? ab[(long long)rc * HW + pxc] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = rbase + (e >> 2) * 8 + (e & 3);
      float v = __uint_as_float(__float_as_uint(acc[i][e]) ^ sflip32) + bv[e] + av[e];
      if (a.epi == PRN_EPI_RELU) v = fmaxf(v, 0.f);
      else if (a.epi == PRN_EPI_SIGMOID) v = 1.f / (1.f + expf(-v));
      if (cok && row < M) yb[(long long)row * HW + px] = v;
    }
  }
}

// ---- the same idea on the fp16 pipe: TWO pieces and THREE products per multiply-add ---------------------------------------------------------
// fp16 carries 11 significand bits, so two pieces hold 22 of fp32's 24 -- if the operand sits inside fp16's narrow exponent range.  Both
// operands are therefore scaled by exact powers of two first: a weight row by 2^(14 - E_a[m]) (E_a = exponent of the row's largest element,
// found by split16_rowmax_kernel), an activation column by 2^(14 - E_b[n]) where E_b is the RUNNING exponent of the column's largest element
// so far -- when a later K slice raises it, the lane's accumulators are rescaled (exact) before the next MFMA; the epilogue undoes both
// scalings.  Per element: h = fp16(x), l = fp16(x - h) (round to nearest: |x - h - l| <= 2^-23 |x|, unbiased; elements more than 2^17 below
// their row's / column's maximum lose relative precision, at an absolute error of 2^-39 of that maximum); products l*h, h*l, h*h -- the
// dropped l*l is <= 2^-22 of the product.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr int IMG16_U4 = 1024;    // uint4 per fp16 image: 2 pieces x 4 k-groups x 128 rows x 16 B = 16 KB

__device__ __forceinline__ void split2_f16(float x0, float x1, unsigned& h, unsigned& l) {
  f16x2_t hv, lv;
  hv[0] = (_Float16)x0; hv[1] = (_Float16)x1;                    // one v_cvt_pk_f16_f32 (round to nearest even)
  h = __builtin_bit_cast(unsigned, hv);
  // residual x - float(h), exact: v_fma_mix_f32 reads the f16 half as its fp32 value (one instruction instead of convert back + subtract)
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(x0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(x1));
  lv[0] = (_Float16)r0; lv[1] = (_Float16)r1;
  l = __builtin_bit_cast(unsigned, lv);
}

// ex[z * Mpad + m] = frexp exponent of max_k |w[z][m][k]| (0 for an all-zero or padding row): one wave per row
__global__ __launch_bounds__(256) void split16_rowmax_kernel(const float* __restrict__ w, int* __restrict__ ex, int M, int K, long long zw, int Mpad, long long rows) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const long long z = row / Mpad; const int m = (int)(row - z * Mpad);
  float mx = 0.f;
  if (m < M) {
    const float* wr = w + z * zw + (long long)m * K;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, fabsf(wr[k]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (lane == 0) ex[row] = mx > 0.f ? __builtin_amdgcn_frexp_expf(mx) : 0;
}

// taps > 1: the images' K axis is TAP-major (k' = t * C + c for the weight w[m][c][t], C = K / taps) -- the order in which the tap-gather kernel walks
// a KH x KW convolution's operand: a 32-deep slice then lies inside one tap and is loaded like a 1x1 layer's, at that tap's pixel shift
__global__ void split16_prepare_kernel(const float* __restrict__ w, uint4* __restrict__ img, const int* __restrict__ ex, int M, int K, long long zw, int mtiles, int kslices,
                                       long long total, int taps = 1) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // one thread per (z, m tile, k slice, k group, row)
  if (i >= total) return;
  const int g = i % 4; const int r = (i / 4) % 128; const long long t = i / 512;      // k group fastest: four lanes read 128 contiguous bytes of one weight row
  const int ks = t % kslices; const long long zm = t / kslices; const int mt = zm % mtiles; const long long z = zm / mtiles;
  const int m = mt * 128 + r, k0 = ks * 32 + g * 8;
  const int sh = 14 - ex[z * (mtiles * 128) + m];
  float v[8];
  const int C = K / taps;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int kp = k0 + j;                                       // position on the images' K axis
    const int ksrc = taps > 1 ? (kp % C) * taps + kp / C : kp;
    v[j] = (m < M && kp < K) ? ldexpf(w[z * zw + (long long)m * K + ksrc], sh) : 0.f;
  }
  uint4 h, l;
  split2_f16(v[0], v[1], h.x, l.x); split2_f16(v[2], v[3], h.y, l.y); split2_f16(v[4], v[5], h.z, l.z); split2_f16(v[6], v[7], h.w, l.w);
  uint4* o = img + t * IMG16_U4;
  o[g * 128 + r] = h; o[(4 + g) * 128 + r] = l;
}

#ifdef PRN_S16_TIMING
// Profiling aid (side build only: tools/split16_phase_timing.py compiles this file with -DPRN_S16_TIMING into its own library): per workgroup,
// the 100 MHz wall clock at kernel entry / loop entry / loop exit / kernel exit and the hardware id (XCC, SE, CU) it ran on.
__device__ long long prn_s16_dbg[8192 * 6];
#define S16_T(slot_) do { if (t == 0 && blockIdx.x + gridDim.x * blockIdx.y < 8192) prn_s16_dbg[(blockIdx.x + gridDim.x * blockIdx.y) * 6 + (slot_)] = (long long)wall_clock64(); } while (0)
__device__ long long prn_s16_it[64 * 4 * 4];      // workgroup 0 (and the one in the middle of the grid), wave 0: per iteration, the clock after the barrier / after the
                                                  // MFMA issue / when the next slice's activations are in registers / at the end of the iteration
#define S16_IT(slot_) do { if (t == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && blockIdx.y == 0 && ks - ks0 < 64) { \
    if ((slot_) == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
    prn_s16_it[((blockIdx.x ? 1 : 0) * 64 + (ks - ks0)) * 4 + (slot_)] = (long long)wall_clock64(); } } while (0)
extern "C" int prn_debug_s16_iter(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(prn_s16_it), sizeof(long long) * (size_t)n); }
extern "C" int prn_debug_s16_timing(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(prn_s16_dbg), sizeof(long long) * (size_t)n); }
#else
#define S16_T(slot_) do { } while (0)
#define S16_IT(slot_) do { } while (0)
#endif

// Store policy of the epilogue (experiment, PRN_SPLIT_STORE_POLICY): 0 plain (write-back in L2: the dirty lines are flushed when the kernel ends),
// 1 sc1 (agent scope: written through), 2 nt, 3 sc0 sc1 (system scope), 4 sc1 nt
__device__ __forceinline__ void store4_policy(float* p, float4 v, int pol) {
  const f32x4_t q = {v.x, v.y, v.z, v.w};
  if (pol == 0) *reinterpret_cast<float4*>(p) = v;
  else if (pol == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(q) : "memory");
  else if (pol == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(q) : "memory");
  else if (pol == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(q) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(q) : "memory");
}

struct Split16Args {
  const uint4* img; const int* ex; const float* x; const float* bias; const float* addend; float* y; float* partial;
  int M, K, B, HW, epi, mtiles, kslices, ptiles, total, splits;
  int wide;                       // epilogue through the LDS transpose (float4 stores): HW % 4 == 0 and 16-byte aligned output / addend / partial planes
  int store_policy;
  long long zx, zy, slice;
  // TAPS (a KH x KW convolution with zero padding as a GEMM over the tap-major K axis of its images): x is [B][C][XH][XW], the GEMM's HW pixels are
  // the Ho x Wo outputs, K = KH * KW * C with C % 32 == 0 -- slice ks is channels (32 ks) % C .. + 31 of tap (32 ks) / C
  int C, XH, XW, Wo, KW, stride, pad;
  // TAPS = 2 (PRN_IN_UP2_PHASE): z = output phase (py, px) with its own [M x 4C] images; the GEMM's pixels are the SOURCE grid, tap (ty, tx) of
  // pixel (i, j) reads source (i + py - 1 + ty, j + px - 1 + tx) clamped to the map (replicate border), the result is stored at (2i + py, 2j + px) of a
  // [B][M][2 XH][2 XW] tensor; no K split, no addend
};

template <int NP, int TAPS = 0>      // NP 3: l*h, h*l, h*h   4: + l*l;  TAPS 1: the activation operand is a zero-padded KH x KW gather, 2: a sub-pixel phase (see Split16Args)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void split16_gemm_kernel(const Split16Args a) {
  __shared__ uint4 lds[2 * IMG16_U4];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int id = prn_xcd_remap(blockIdx.x, a.total);
  const int mt = id % a.mtiles; const int rest = id / a.mtiles;
  const int pt = rest % a.ptiles; const int zb = rest / a.ptiles; const int b = zb % a.B, z = zb / a.B;
  const int sp = blockIdx.y;
  const int ks0 = (int)((long long)a.kslices * sp / a.splits), ks1 = (int)((long long)a.kslices * (sp + 1) / a.splits);
  const int HW = a.HW, M = a.M;
  const int r = lane & 31, gs = lane >> 5;
  S16_T(0);
#ifdef PRN_S16_TIMING
  if (t == 0 && blockIdx.x + gridDim.x * blockIdx.y < 8192) {
    unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    prn_s16_dbg[(blockIdx.x + gridDim.x * blockIdx.y) * 6 + 4] = ((long long)(xcc & 15) << 32) | hw;
  }
#endif
  const int px = pt * 128 + wave * 32 + r;
  const int pxc = px < HW ? px : HW - 1;
  const int XHW = TAPS ? a.XH * a.XW : HW;                         // elements of one activation channel plane
  const int XK = TAPS ? a.C : a.K;                                 // channel planes of one image
  const float* xb = a.x + (long long)z * a.zx + (long long)b * XK * XHW;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, XK * XHW * 4, 0x00020000);
  const int xoff = (pxc + gs * 8 * HW) * 4;
  int choff[16];                                                   // scalar byte offsets of a lane's 16 channels within a slice, clamped to the buffer: the range check
#pragma unroll                                                     // subtracts the scalar offset from the size, which must not wrap (K < 32)
  for (int j = 0; j < 16; ++j) choff[j] = __builtin_amdgcn_readfirstlane(min(((j >> 3) * 16 + (j & 7)) * XHW * 4, XK * XHW * 4));
  // TAPS: top-left input position of this lane's output pixel; a slice's byte offset is that of its tap (or the out-of-range marker: zero padding)
  const int t_oi = TAPS ? pxc / a.Wo : 0, t_oj = TAPS ? pxc - t_oi * a.Wo : 0;
  const int t_iy0 = t_oi * a.stride - a.pad, t_ix0 = t_oj * a.stride - a.pad;
  auto tap_vo = [&](int ks_) -> int {
    const int k0 = ks_ * BK, tp = __builtin_amdgcn_readfirstlane(k0 / a.C), c0 = __builtin_amdgcn_readfirstlane(k0 - tp * a.C);
    const int ty = __builtin_amdgcn_readfirstlane(tp / a.KW), tx = __builtin_amdgcn_readfirstlane(tp - ty * a.KW);
    int iy = t_iy0 + ty, ix = t_ix0 + tx;
    bool ok = (unsigned)iy < (unsigned)a.XH && (unsigned)ix < (unsigned)a.XW && k0 < a.K;
    if (TAPS == 2) {
      iy = min(max(t_oi + (z >> 1) - 1 + ty, 0), a.XH - 1); ix = min(max(t_oj + (z & 1) - 1 + tx, 0), a.XW - 1);
      ok = k0 < a.K;
    }
    return ok ? (iy * a.XW + ix + (c0 + gs * 8) * XHW) * 4 : (int)0x80000000u;
  };
  const uint4* ag = a.img + ((long long)(z * a.mtiles + mt) * a.kslices) * IMG16_U4;
  const i32x4_t adesc = make_desc(ag, (unsigned)a.kslices * IMG16_U4 * 16u);
  const unsigned lds0 = (unsigned)(unsigned long long)(void*)lds;
  float rn[16];
  f16x8_t bp[2][2];
  int erun = -1000, de = 0;        // running exponent of this column's largest element; pending rescale of the accumulators
  // The 16-bit pipe's accumulate step truncates toward -infinity: every output carries an error of about -1e-9 of sum|w||x| whatever its sign --
  // 1/20 of the rounding noise per element, but COHERENT, so it adds up linearly in every later sum over pixels (BatchNorm / GroupNorm
  // statistics and their backward sums, weight gradients) where rounding noise adds up as a square root.  Odd pixels therefore enter with
  // their sign flipped (the pieces' sign bits; the epilogue flips the result back): the truncation error then alternates in sign from one
  // pixel to the next and cancels in spatial sums.  Costs 16 v_xor per slice.  Which pixels: those whose index has an odd number of set bits
  // (the Thue-Morse sequence), not simply the odd ones -- a plain alternation along x survives every stride-2 subsampling (downsample
  // convolutions, resize x0.5) as a constant sign; t(2n) = t(n) makes the subsampled pattern the same balanced sequence again.
  const unsigned podd = (unsigned)__builtin_popcount((unsigned)px) & 1u;
  const unsigned sflip16 = podd ? 0x80008000u : 0u, sflip32 = podd ? 0x80000000u : 0u;
#define S16_DMA(ks_, st_) do { \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) \
      lds_dma16(lds0 + (unsigned)(st_) * (IMG16_U4 * 16) + (unsigned)(i * 4 + wave) * 1024u, adesc, (unsigned)lane * 16u, ((ks_) * IMG16_U4 + (i * 4 + wave) * 64) * 16); \
  } while (0)
#define S16_LOADB(ks_) do { \
    const int vo = TAPS ? tap_vo(ks_) : xoff + (ks_) * (BK * 4) * HW;   /* one vector add per slice; the 16 channel offsets are scalars */ \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) \
        rn[s2 * 8 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vo, choff[s2 * 8 + j], 0)); \
  } while (0)
#define S16_PIECES() do { \
    float mx = 0.f; \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) mx = fmaxf(mx, fabsf(rn[j])); \
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                       /* the two lanes of a column hold different k */ \
    const int en = max(erun, mx > 0.f ? __builtin_amdgcn_frexp_expf(mx) : -200); \
    de = erun - en; erun = en; \
    const int sh = 14 - en; \
    if (__builtin_amdgcn_ballot_w64(sh > 127 && mx > 0.f) == 0ull) { \
      /* scale and sign in one packed multiply by +-2^sh (exact; a column of zeros so far may carry any scale) */ \
      f32x2_t sc; sc[0] = __uint_as_float(((unsigned)(min(sh, 127) + 127) << 23) | sflip32); sc[1] = sc[0]; \
      _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) { \
        uint4 h, l; f32x2_t v; \
        v[0] = rn[s2 * 8 + 0]; v[1] = rn[s2 * 8 + 1]; v = v * sc; split2_f16(v[0], v[1], h.x, l.x); \
        v[0] = rn[s2 * 8 + 2]; v[1] = rn[s2 * 8 + 3]; v = v * sc; split2_f16(v[0], v[1], h.y, l.y); \
        v[0] = rn[s2 * 8 + 4]; v[1] = rn[s2 * 8 + 5]; v = v * sc; split2_f16(v[0], v[1], h.z, l.z); \
        v[0] = rn[s2 * 8 + 6]; v[1] = rn[s2 * 8 + 7]; v = v * sc; split2_f16(v[0], v[1], h.w, l.w); \
        bp[s2][0] = __builtin_bit_cast(f16x8_t, h); bp[s2][1] = __builtin_bit_cast(f16x8_t, l); \
      } \
    } else {                                                      /* activations below 2^-113: 2^sh is not a float */ \
      _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) { \
        uint4 h, l; \
        split2_f16(ldexpf(rn[s2 * 8 + 0], sh), ldexpf(rn[s2 * 8 + 1], sh), h.x, l.x); split2_f16(ldexpf(rn[s2 * 8 + 2], sh), ldexpf(rn[s2 * 8 + 3], sh), h.y, l.y); \
        split2_f16(ldexpf(rn[s2 * 8 + 4], sh), ldexpf(rn[s2 * 8 + 5], sh), h.z, l.z); split2_f16(ldexpf(rn[s2 * 8 + 6], sh), ldexpf(rn[s2 * 8 + 7], sh), h.w, l.w); \
        h.x ^= sflip16; h.y ^= sflip16; h.z ^= sflip16; h.w ^= sflip16; l.x ^= sflip16; l.y ^= sflip16; l.z ^= sflip16; l.w ^= sflip16; \
        bp[s2][0] = __builtin_bit_cast(f16x8_t, h); bp[s2][1] = __builtin_bit_cast(f16x8_t, l); \
      } \
    } } while (0)
#define S16_STEP(s2_) do { \
    const int kg = 2 * (s2_) + gs; \
    const f16x8_t bh = bp[s2_][0], bl = bp[s2_][1]; \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { \
      const uint4* ap = lds + st * IMG16_U4 + kg * 128 + i * 32 + r; \
      const f16x8_t ah = __builtin_bit_cast(f16x8_t, ap[0]), al = __builtin_bit_cast(f16x8_t, ap[512]); \
      f32x16_t c = acc[i]; \
      if (NP >= 4) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, c, 0, 0, 0); \
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0); \
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0); \
      acc[i] = c; \
    } } while (0)
  f32x16_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  // Schedule (round 4): the activation loads of slice ks + 2 are issued at the END of iteration ks, right after the registers they land in
  // were cut into the pieces of slice ks + 1 -- so they have a whole iteration (barrier, weight DMA, 24 MFMAs, the cut) to arrive, at no
  // register cost.  Before, they were issued at the top of iteration ks + 1 and needed at its end: only the MFMA phase covered their latency.
  S16_DMA(ks0, 0);
  S16_LOADB(ks0);
  S16_PIECES();
  de = 0;                                                        // nothing accumulated yet
  S16_LOADB(ks0 + 1);                                            // (slices past K read zeros through the descriptor's range check)
  S16_T(1);
  for (int ks = ks0; ks < ks1; ++ks) {
    const int st = (ks - ks0) & 1;
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");           // this slice's weight DMA is older than the 16 activation loads that stay in flight
    __syncthreads();
    if (__builtin_amdgcn_ballot_w64(de != 0) != 0ull) {          // a column's maximum grew: bring its partial sums to the new scale (exact)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = ldexpf(acc[i][e], de);
    }
    const bool more = ks + 1 < ks1;
    S16_IT(0);
    if (more) S16_DMA(ks + 1, st ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    S16_STEP(0);
    __builtin_amdgcn_sched_barrier(0);
    S16_STEP(1);
    S16_IT(1);
    S16_IT(2);
    if (more) S16_PIECES(); else de = 0;                        // slice ks + 1 (of THIS K split only: the running scale must not see the next split's data)
    __builtin_amdgcn_sched_barrier(0);
    S16_LOADB(ks + 2);
    S16_IT(3);
  }
#undef S16_DMA
#undef S16_LOADB
#undef S16_PIECES
#undef S16_STEP
  S16_T(2);
  const bool cok = px < HW;
  const int* exm = a.ex + (long long)z * a.mtiles * 128 + mt * 128;
  float* yb = a.y + (long long)z * a.zy + (long long)b * M * HW;
  const float* ab = a.addend ? a.addend + (long long)z * a.zy + (long long)b * M * HW : nullptr;
  float* pb = a.splits > 1 ? a.partial + (long long)sp * a.slice + ((long long)z * a.B + b) * (long long)M * HW : nullptr;
  if (a.wide) {
    // Epilogue through an LDS transpose (the weight images are dead by now): an accumulator block holds 16 rows of ONE pixel per lane, i.e. 16
    // dword stores per block and lane; transposed, a lane owns four consecutive pixels of one row -- 4 dwordx4 stores, bias / addend as float4s.
    // (Host side: HW % 4 == 0 and 16-byte aligned y / addend / partial planes.)
    __syncthreads();                                               // every wave has read its last weight fragments
    float* cw = reinterpret_cast<float*>(lds) + wave * (32 * 36);  // 32 rows x 36 floats per wave
    int* exl = reinterpret_cast<int*>(lds) + 4 * (32 * 36);        // the tile's 128 row exponents: one global load per row instead of one per accumulator element
    if (t < 128) exl[t] = exm[t];
    __syncthreads();
    const int crow = lane >> 3, ccol = (lane & 7) * 4;
    const int px4 = pt * 128 + wave * 32 + ccol;
    const bool pok = px4 < HW;
    const bool fin = a.splits == 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ro = (e >> 2) * 8 + (e & 3) + gs * 4;
        cw[ro * 36 + r] = __uint_as_float(__float_as_uint(ldexpf(acc[i][e], erun + exl[i * 32 + ro] - 28)) ^ sflip32);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rl = crow + 8 * q, row = mt * BM + i * 32 + rl;
        float4 v = *reinterpret_cast<const float4*>(&cw[rl * 36 + ccol]);
        if (!pok || row >= M) continue;
        const long long idx = (long long)row * HW + px4;
        if (TAPS == 2) {                                             // phase-interleaved store: four source pixels of one row -> every second output pixel
          if (a.bias) { const float bm = a.bias[row]; v.x += bm; v.y += bm; v.z += bm; v.w += bm; }
          if (a.epi == PRN_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          const int oi4 = px4 / a.Wo, oj4 = px4 - oi4 * a.Wo;
          float* yp = a.y + (((long long)b * M + row) * (2 * a.XH) + 2 * oi4 + (z >> 1)) * (2 * a.XW) + 2 * oj4 + (z & 1);
          yp[0] = v.x; yp[2] = v.y; yp[4] = v.z; yp[6] = v.w;
          continue;
        }
        if (fin) {
          if (a.bias) { const float bm = a.bias[row]; v.x += bm; v.y += bm; v.z += bm; v.w += bm; }
          if (ab) { const float4 t4 = *reinterpret_cast<const float4*>(ab + idx); v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w; }
          if (a.epi == PRN_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          else if (a.epi == PRN_EPI_SIGMOID) { v.x = 1.f / (1.f + expf(-v.x)); v.y = 1.f / (1.f + expf(-v.y)); v.z = 1.f / (1.f + expf(-v.z)); v.w = 1.f / (1.f + expf(-v.w)); }
          store4_policy(yb + idx, v, a.store_policy);
        } else {
          store4_policy(pb + idx, v, a.store_policy);
        }
      }
      __builtin_amdgcn_wave_barrier();                             // the pad is rewritten by the next block
    }
#ifdef PRN_S16_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    S16_T(3);
    return;
  }
  if (a.splits > 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rl = i * 32 + gs * 4 + (e >> 2) * 8 + (e & 3), row = mt * BM + rl;
        if (cok && row < M) pb[(long long)row * HW + px] = __uint_as_float(__float_as_uint(ldexpf(acc[i][e], erun + exm[rl] - 28)) ^ sflip32);
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rl0 = i * 32 + gs * 4, rbase = mt * BM + rl0;
    float bv[16], av[16]; int ev[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ro = (e >> 2) * 8 + (e & 3), row = rbase + ro; const int rc = row < M ? row : M - 1;
      bv[e] = a.bias ? a.bias[rc] : 0.f; av[e] = ab ? ab[(long long)rc * HW + pxc] : 0.f; ev[e] = exm[rl0 + ro];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = rbase + (e >> 2) * 8 + (e & 3);
      float v = __uint_as_float(__float_as_uint(ldexpf(acc[i][e], erun + ev[e] - 28)) ^ sflip32) + bv[e] + av[e];
      if (a.epi == PRN_EPI_RELU) v = fmaxf(v, 0.f);
      else if (a.epi == PRN_EPI_SIGMOID) v = 1.f / (1.f + expf(-v));
      if (cok && row < M) yb[(long long)row * HW + px] = v;
    }
  }
}

// one launch for many weights: item i covers blocks [first_i, first_{i+1}) of 256 threads = 256 (z, m tile, k slice, k group, row) tuples
struct PrepItem { const float* src; uint4* dst; int M, K, nz, pad; long long zw; long long first; };   // zw: first block of four rows (fp16 row pass); weights are dense
__global__ void split_prepare_batched_kernel(const PrepItem* __restrict__ items, int n) {
  int lo = 0, hi = n - 1;
  const long long blk = blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (items[mid].first <= blk) lo = mid; else hi = mid - 1; }
  const PrepItem it = items[lo];
  const int mtiles = (it.M + 127) / 128, kslices = (it.K + 31) / 32;
  const long long total = (long long)it.nz * mtiles * kslices * 512;
  const long long i = (blk - it.first) * 256 + threadIdx.x;
  if (i >= total) return;
  const int g = i % 4; const int r = (i / 4) % 128; const long long t = i / 512;      // k group fastest: four lanes read 128 contiguous bytes of one weight row
  const int ks = t % kslices; const long long zm = t / kslices; const int mt = zm % mtiles; const long long z = zm / mtiles;
  const int m = mt * 128 + r, k0 = ks * 32 + g * 8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (m < it.M && k0 + j < it.K) ? it.src[z * (long long)it.M * it.K + (long long)m * it.K + k0 + j] : 0.f;
  uint4 h, mm, l;
  split2(v[0], v[1], h.x, mm.x, l.x); split2(v[2], v[3], h.y, mm.y, l.y); split2(v[4], v[5], h.z, mm.z, l.z); split2(v[6], v[7], h.w, mm.w, l.w);
  uint4* o = it.dst + t * IMG_U4;
  o[(0 * 4 + g) * 128 + r] = h; o[(1 * 4 + g) * 128 + r] = mm; o[(2 * 4 + g) * 128 + r] = l;
}

// the fp16 piece format of the same batch: row exponents first (one wave per row of every item), then the images
__device__ __forceinline__ PrepItem find_item(const PrepItem* items, int n, long long blk, bool rows) {
  int lo = 0, hi = n - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((rows ? items[mid].zw : items[mid].first) <= blk) lo = mid; else hi = mid - 1; }
  return items[lo];
}
// (rows pass: the item's `zw` field holds its first block of FOUR rows -- a second table, see prn_split_prepare_batched)
__global__ __launch_bounds__(256) void split16_rowmax_batched_kernel(const PrepItem* __restrict__ items, int n) {
  const PrepItem it = find_item(items, n, blockIdx.x, true);
  const int mtiles = (it.M + 127) / 128, Mpad = mtiles * 128;
  const long long rows = (long long)it.nz * Mpad;
  const long long row = ((long long)blockIdx.x - it.zw) * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int kslices = (it.K + 31) / 32;
  int* ex = (int*)((char*)it.dst + (long long)it.nz * mtiles * kslices * IMG16_U4 * 16);
  const int lane = threadIdx.x & 63;
  const long long z = row / Mpad; const int m = (int)(row - z * Mpad);
  float mx = 0.f;
  if (m < it.M) {
    const float* wr = it.src + z * (long long)it.M * it.K + (long long)m * it.K;
    if ((it.K & 3) == 0 && (reinterpret_cast<uintptr_t>(it.src) & 15) == 0) {       // rows are 16-byte aligned: 16 bytes per lane and load (a 256-deep row is one instruction)
      const float4* w4 = reinterpret_cast<const float4*>(wr);
      for (int k = lane; k < (it.K >> 2); k += 64) { const float4 q = w4[k]; mx = fmaxf(fmaxf(mx, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w))); }
    } else {
      for (int k = lane; k < it.K; k += 64) mx = fmaxf(mx, fabsf(wr[k]));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (lane == 0) ex[row] = mx > 0.f ? __builtin_amdgcn_frexp_expf(mx) : 0;
}
__global__ void split16_prepare_batched_kernel(const PrepItem* __restrict__ items, int n) {
  const PrepItem it = find_item(items, n, blockIdx.x, false);
  const int mtiles = (it.M + 127) / 128, kslices = (it.K + 31) / 32;
  const long long total = (long long)it.nz * mtiles * kslices * 512;
  const long long i = ((long long)blockIdx.x - it.first) * 256 + threadIdx.x;
  if (i >= total) return;
  const int* ex = (const int*)((const char*)it.dst + (long long)it.nz * mtiles * kslices * IMG16_U4 * 16);
  const int g = i % 4; const int r = (i / 4) % 128; const long long t = i / 512;      // k group fastest: four lanes read 128 contiguous bytes of one weight row
  const int ks = t % kslices; const long long zm = t / kslices; const int mt = zm % mtiles; const long long z = zm / mtiles;
  const int m = mt * 128 + r, k0 = ks * 32 + g * 8;
  const int sh = 14 - ex[z * (mtiles * 128) + m];
  float v[8];
  const float* wr = it.src + z * (long long)it.M * it.K + (long long)m * it.K + k0;
  if (m < it.M && k0 + 8 <= it.K && (it.K & 3) == 0 && (reinterpret_cast<uintptr_t>(it.src) & 15) == 0) {     // the 32 bytes of this thread as two 16-byte loads
    const float4 a = *reinterpret_cast<const float4*>(wr), b = *reinterpret_cast<const float4*>(wr + 4);
    v[0] = ldexpf(a.x, sh); v[1] = ldexpf(a.y, sh); v[2] = ldexpf(a.z, sh); v[3] = ldexpf(a.w, sh);
    v[4] = ldexpf(b.x, sh); v[5] = ldexpf(b.y, sh); v[6] = ldexpf(b.z, sh); v[7] = ldexpf(b.w, sh);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (m < it.M && k0 + j < it.K) ? ldexpf(wr[j], sh) : 0.f;
  }
  uint4 h, l;
  split2_f16(v[0], v[1], h.x, l.x); split2_f16(v[2], v[3], h.y, l.y); split2_f16(v[4], v[5], h.z, l.z); split2_f16(v[6], v[7], h.w, l.w);
  uint4* o = it.dst + t * IMG16_U4;
  o[g * 128 + r] = h; o[(4 + g) * 128 + r] = l;
}

}  // namespace

// ---- internal interface (prn_common.h) ------------------------------------------------------------------------------------------------
// Should y[nz][B][M][HW] = w[nz][M][K] * x[nz][B][K][HW] run on the split kernel under `o`, and with how many K splits?  0 = no.
// Plan (tools/native/gemm_split_lab on the step's shapes): the kernel wins where at least ~300 of its 128 x 128 tiles exist and the last
// m tile is more than half full; short of tiles, a K split of 2 .. 3 fills the GPU as long as every split keeps >= 8 slices.
// The threshold (split_min_tiles) is a BOARD-level trade the caller makes: broad use of the 16-bit pipe makes the firmware lower the
// shader clock for everything else (bf16 pieces: 2.35 -> 2.17 GHz over a training step, fp16 pieces: -> 2.25 GHz).
int prn_split_gemm_plan(int M, int K, int B, int HW, int nz, const prn_gemm_opts* o) {
  if (o == nullptr || o->split_mode == PRN_SPLIT_OFF) return 0;
  if (M <= 0 || K <= 0 || B <= 0 || HW <= 0 || nz <= 0) return 0;
  if ((int64_t)K * HW >= (1LL << 29) || (int64_t)M * HW >= (1LL << 29)) return 0;
  const int mtiles = cdiv(M, 128), kslices = cdiv(K, 32);
  const int64_t tiles = (int64_t)mtiles * cdiv(HW, 128) * B * nz;
  int splits = 1;
  if (tiles < 300) {
    splits = (int)(640 / (tiles > 0 ? tiles : 1));
    if (splits > kslices / 8) splits = kslices / 8;
    static const int smax = prn_env_int("PRN_SPLIT_KSPLIT_MAX", 3);                                        // PRN_SPLIT_KSPLIT_MAX (tuning): 3.  Round 4: training step 45.4 (2) / 44.0 (3) / 44.1 (4) ms; since the partial sums of     // the stage-3 layers are summed by the BatchNorm kernel that reads them (round 5: one partial tensor less to read there) 43.51 / 43.52 (4) -> 43.35 / 43.30 (3), 43.40 / 43.50 (2).
                                                                 // A performance choice only (the fixture's ReLU-boundary channel that 3 splits once tipped is bounded separately by the parity test since round 5)
    // ... but a launch that reaches the tile floor only with four splits keeps four (the stage-4 reducing layers, 96 tiles: 288 < 300 with three)
    const int wide = splits > 4 ? 4 : splits;
    if (splits > smax) splits = smax;
    if (o->split_mode == PRN_SPLIT_PLAN && tiles * splits < o->split_min_tiles && tiles * wide >= o->split_min_tiles) splits = wide;
    if (splits < 1) splits = 1;
  }
  if (o->split_mode == PRN_SPLIT_ALWAYS) return splits;
  if ((int64_t)mtiles * 128 * 4 > (int64_t)M * 5) return 0;      // more than a fifth of the row tiles' rows would be padding (M = 64, 160, 192: yes; 3728: no)
  if (2.0 * M * K * (double)HW * B * nz < (double)o->split_min_gflop * 1e9) return 0;   // small launches are all launch latency: one kernel beats split + GEMM (+ sum)
  if (tiles * splits < o->split_min_tiles) return 0;
  return splits;
}
extern "C" void prn_gemm_opts_default(prn_gemm_opts* o) {
  if (!o) return;
  o->split_mode = PRN_SPLIT_PLAN; o->split_kind = PRN_PIECES_F16; o->split_products = 3; o->split_min_tiles = 300; o->split_min_gflop = 4.0f;
  o->wgrad_wgs = 0; o->wgrad_target = 0; o->wgrad_split = PRN_SPLIT_PLAN;
}
extern "C" int64_t prn_split_images_bytes(int M, int K, int nz) {
  if (M <= 0 || K <= 0 || nz <= 0) return -1;
  return prn_split_gemm_image_bytes(M, K, nz);
}
extern "C" int prn_split_prepare_batched(const void* items_dev, int n_items, int64_t total_blocks, int64_t total_row_blocks, int kind, void* stream) {
  PRN_REQUIRE(items_dev && n_items > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "prn_split_prepare_batched: bad arguments");
  PRN_REQUIRE(kind == PRN_PIECES_F16 || kind == PRN_PIECES_BF16, "prn_split_prepare_batched: kind is PRN_PIECES_F16 or PRN_PIECES_BF16");
  if (kind == PRN_PIECES_F16) {
    PRN_REQUIRE(total_row_blocks > 0 && total_row_blocks < (1LL << 31), "prn_split_prepare_batched: the fp16 piece format needs the row-block count");
    hipLaunchKernelGGL(split16_rowmax_batched_kernel, dim3((unsigned)total_row_blocks), dim3(256), 0, (hipStream_t)stream, (const PrepItem*)items_dev, n_items);
    hipLaunchKernelGGL(split16_prepare_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const PrepItem*)items_dev, n_items);
    PRN_CHECK_LAUNCH("prn_split_prepare_batched/f16");
    return 0;
  }
  hipLaunchKernelGGL(split_prepare_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const PrepItem*)items_dev, n_items);
  PRN_CHECK_LAUNCH("prn_split_prepare_batched");
  return 0;
}
namespace {
// cuts w[nz][M][K] (z stride zw) into `images` in the given piece format
int cut_weight(const float* w, void* images, int M, int K, int nz, long long zw, int kind, hipStream_t st, const char* who, int taps = 1) {
  const int mtiles = cdiv(M, 128), kslices = cdiv(K, 32);
  const long long ptotal = (long long)nz * mtiles * kslices * 512;
  if (kind == PRN_PIECES_F16) {
    int* ex = (int*)((char*)images + (int64_t)nz * mtiles * kslices * IMG16_U4 * 16);
    const long long rows = (long long)nz * mtiles * 128;
    hipLaunchKernelGGL(split16_rowmax_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, w, ex, M, K, zw, mtiles * 128, rows);
    hipLaunchKernelGGL(split16_prepare_kernel, dim3(cdiv(ptotal, 256)), dim3(256), 0, st, w, (uint4*)images, (const int*)ex, M, K, zw, mtiles, kslices, ptotal, taps);
  } else {
    hipLaunchKernelGGL(split_prepare_kernel, dim3(cdiv(ptotal, 256)), dim3(256), 0, st, w, (uint4*)images, M, K, zw, mtiles, kslices, ptotal);
  }
  PRN_CHECK_LAUNCH(who);
  return 0;
}
}  // namespace
// cuts ONE dense weight [nz][M][K] into images of the given piece format (prn_split_images_bytes bytes)
extern "C" int prn_split_prepare(const float* w, void* images, int M, int K, int nz, int kind, void* stream) {
  PRN_REQUIRE(w && images && M > 0 && K > 0 && nz > 0 && (reinterpret_cast<uintptr_t>(images) & 15) == 0, "prn_split_prepare: bad arguments");
  PRN_REQUIRE(kind == PRN_PIECES_F16 || kind == PRN_PIECES_BF16, "prn_split_prepare: kind is PRN_PIECES_F16 or PRN_PIECES_BF16");
  return cut_weight(w, images, M, K, nz, (long long)M * K, kind, (hipStream_t)stream, "prn_split_prepare");
}
extern "C" int prn_gemm_pipe(int M, int K, int B, int HW, int nz, const prn_gemm_opts* opts) {
  return prn_split_gemm_plan(M, K, B, HW, nz, opts);
}
// (the larger of the two kinds' images, plus the fp16 kind's row exponents behind them)
int64_t prn_split_gemm_image_bytes(int M, int K, int nz) { return (int64_t)nz * cdiv(M, 128) * cdiv(K, 32) * IMG_U4 * 16 + (int64_t)nz * cdiv(M, 128) * 128 * 4; }
int64_t prn_split_gemm_partial_bytes(int M, int B, int HW, int nz, int splits) { return splits > 1 ? (int64_t)splits * nz * B * M * HW * 4 : 0; }

// w_images: current images of w handed in by the caller (nullptr: cut w into images_ws first).  partial: prn_split_gemm_partial_bytes
// (splits > 1).  zw / zx / zy: element strides per z.
// phase: 0 = everything, 1 = the split + GEMM launches only, 2 = the K-split sum only (profiler brackets, like prn_conv2d_fwd_phase).
int prn_split_gemm(const float* w, const void* w_images, const float* x, const float* bias, const float* addend, float* y, void* images_ws, float* partial, int M,
                   int K, int B, int HW, int nz, int64_t zw, int64_t zx, int64_t zy, int epi, int splits, const prn_gemm_opts* o, hipStream_t st, int phase) {
  PRN_REQUIRE(o != nullptr && w && x && y && (splits == 1 || partial), "prn_split_gemm: null operand");
  const int mtiles = cdiv(M, 128), kslices = cdiv(K, 32), ptiles = cdiv(HW, 128);
  PRN_REQUIRE((int64_t)kslices * IMG_U4 * 16 < (1LL << 31) && (int64_t)K * HW < (1LL << 29), "prn_split_gemm: operand larger than a buffer descriptor");
  const int kind = o->split_kind;
  PRN_REQUIRE(kind == PRN_PIECES_F16 || kind == PRN_PIECES_BF16, "prn_split_gemm: unknown piece format %d", kind);
  {
    static const int dbg = prn_env_int("PRN_SPLIT_DEBUG", 0);                                           // PRN_SPLIT_DEBUG=1: one line per launch on stderr (which shapes a plan puts on the kernel)
    if (dbg) fprintf(stderr, "prn_split_gemm M=%d K=%d B=%d HW=%d nz=%d splits=%d epi=%d addend=%d\n", M, K, B, HW, nz, splits, epi, addend != nullptr);
  }
  if (phase != 2) {
    const void* images = w_images;
    if (images == nullptr) {
      PRN_REQUIRE(images_ws != nullptr, "prn_split_gemm: no workspace for the weight images");
      if (int e = cut_weight(w, images_ws, M, K, nz, (long long)zw, kind, st, "prn_split_gemm/prepare")) return e;
      images = images_ws;
    }
    PRN_REQUIRE((reinterpret_cast<uintptr_t>(images) & 15) == 0, "prn_split_gemm: images must be 16-byte aligned");
    if (kind == PRN_PIECES_F16) {
      Split16Args a;
      a.img = (const uint4*)images; a.ex = (const int*)((const char*)images + (int64_t)nz * mtiles * kslices * IMG16_U4 * 16);
      a.x = x; a.bias = bias; a.addend = addend; a.y = y; a.partial = partial;
      a.M = M; a.K = K; a.B = B; a.HW = HW; a.epi = epi; a.mtiles = mtiles; a.kslices = kslices; a.ptiles = ptiles;
      a.total = mtiles * ptiles * B * nz; a.splits = splits; a.zx = zx; a.zy = zy; a.slice = (long long)nz * B * M * HW;
      a.C = K; a.XH = 1; a.XW = HW; a.Wo = HW; a.KW = 1; a.stride = 1; a.pad = 0;
      {
        auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        static const int wide_on = prn_env_int("PRN_SPLIT_WIDE_STORE", 1);                                 // PRN_SPLIT_WIDE_STORE=0: the per-element epilogue (A/B)
        a.wide = wide_on && (HW & 3) == 0 && ((int64_t)M * HW & 3) == 0 && (zy & 3) == 0 && al16(y) && al16(addend) && al16(partial);
        static const int policy = prn_env_int("PRN_SPLIT_STORE_POLICY", 1);
        a.store_policy = policy;
      }
      if (o->split_products >= 4) hipLaunchKernelGGL(split16_gemm_kernel<4>, dim3(a.total, splits), dim3(256), 0, st, a);
      else hipLaunchKernelGGL(split16_gemm_kernel<3>, dim3(a.total, splits), dim3(256), 0, st, a);
      PRN_CHECK_LAUNCH("prn_split_gemm/f16");
    } else {
      SplitArgs a;
      a.img = (const uint4*)images; a.x = x; a.bias = bias; a.addend = addend; a.y = y; a.partial = partial;
      a.M = M; a.K = K; a.B = B; a.HW = HW; a.epi = epi; a.mtiles = mtiles; a.kslices = kslices; a.ptiles = ptiles;
      a.total = mtiles * ptiles * B * nz; a.splits = splits; a.zx = zx; a.zy = zy; a.slice = (long long)nz * B * M * HW;
      hipLaunchKernelGGL(split_gemm_kernel, dim3(a.total, splits), dim3(256), 0, st, a);
      PRN_CHECK_LAUNCH("prn_split_gemm");
    }
  }
  if (splits > 1 && phase != 1) {
    PRN_REQUIRE(nz == 1 || (zy == (int64_t)B * M * HW), "prn_split_gemm: K splits need a dense output");
    return prn_launch_reduce_epilogue(partial, bias, addend, y, (int64_t)nz * B * M * HW, M, HW, splits, epi, st);
  }
  return 0;
}

// A zero-padded KH x KW convolution y[b][m][oh][ow] = sum_{c,r,s} w[m][c][r][s] x[b][c][oh * stride - pad + r][ow * stride - pad + s] on the 16-bit pipe
// (fp16 pieces only): the GEMM of prn_split_gemm over a TAP-major K axis (C % 32 == 0: every 32-deep slice is 32 channels of one tap, fetched like a
// 1x1 layer's slice at that tap's pixel shift, out-of-image positions through the descriptor's range check).  The weight is cut per call into
// images_ws (prn_split_gemm_image_bytes(M, KH * KW * C, 1)); partial: prn_split_gemm_partial_bytes(M, B, Ho * Wo, 1, splits).
// Used for the 4x4 / stride-2 input gradient of the depth decoder's sub-pixel upsample-convolutions (planerecnet.py:540-566).
int prn_split_conv_taps(const float* w, const float* x, const float* bias, const float* addend, float* y, void* images_ws, float* partial, int M, int C, int B, int XH,
                        int XW, int Ho, int Wo, int KH, int KW, int stride, int pad, int epi, int splits, const prn_gemm_opts* o, hipStream_t st, int phase) {
  PRN_REQUIRE(o != nullptr && w && x && y && images_ws && (splits == 1 || partial), "prn_split_conv_taps: null operand");
  PRN_REQUIRE(o->split_kind == PRN_PIECES_F16 && (C & 31) == 0 && KH > 0 && KW > 0, "prn_split_conv_taps: fp16 pieces and C %% 32 == 0 only");
  const int K = KH * KW * C, HW = Ho * Wo;
  const int mtiles = cdiv(M, 128), kslices = K / 32, ptiles = cdiv(HW, 128);
  PRN_REQUIRE((int64_t)kslices * IMG_U4 * 16 < (1LL << 31) && (int64_t)C * XH * XW < (1LL << 29) && (int64_t)M * HW < (1LL << 29),
              "prn_split_conv_taps: operand larger than a buffer descriptor");
  if (phase != 2) {
    if (int e = cut_weight(w, images_ws, M, K, 1, (long long)M * K, PRN_PIECES_F16, st, "prn_split_conv_taps/prepare", KH * KW)) return e;
    Split16Args a;
    a.img = (const uint4*)images_ws; a.ex = (const int*)((const char*)images_ws + (int64_t)mtiles * kslices * IMG16_U4 * 16);
    a.x = x; a.bias = bias; a.addend = addend; a.y = y; a.partial = partial;
    a.M = M; a.K = K; a.B = B; a.HW = HW; a.epi = epi; a.mtiles = mtiles; a.kslices = kslices; a.ptiles = ptiles;
    a.total = mtiles * ptiles * B; a.splits = splits; a.zx = 0; a.zy = 0; a.slice = (long long)B * M * HW;
    a.C = C; a.XH = XH; a.XW = XW; a.Wo = Wo; a.KW = KW; a.stride = stride; a.pad = pad;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    a.wide = (HW & 3) == 0 && ((int64_t)M * HW & 3) == 0 && al16(y) && al16(addend) && al16(partial);
    a.store_policy = 1;
    if (o->split_products >= 4) hipLaunchKernelGGL((split16_gemm_kernel<4, 1>), dim3(a.total, splits), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((split16_gemm_kernel<3, 1>), dim3(a.total, splits), dim3(256), 0, st, a);
    PRN_CHECK_LAUNCH("prn_split_conv_taps");
  }
  if (splits > 1 && phase != 1) return prn_launch_reduce_epilogue(partial, bias, addend, y, (int64_t)B * M * HW, M, HW, splits, epi, st);
  return 0;
}

// Upsample(x2, nearest) -> ReflectionPad2d(1) -> Conv3x3 in its sub-pixel form (PRN_IN_UP2_PHASE: wp [4][M][C][2][2], y [B][M][2H][2W]) on the 16-bit pipe:
// the four phases are the z axis of ONE launch (tap-major images per phase, cut per call into images_ws: prn_split_gemm_image_bytes(M, 4 C, 4)).
int prn_split_conv_up2(const float* wp, const float* x, const float* bias, float* y, void* images_ws, int M, int C, int B, int H, int W, int epi, const prn_gemm_opts* o,
                       hipStream_t st) {
  PRN_REQUIRE(o != nullptr && wp && x && y && images_ws, "prn_split_conv_up2: null operand");
  PRN_REQUIRE(o->split_kind == PRN_PIECES_F16 && (C & 31) == 0 && (W & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0, "prn_split_conv_up2: fp16 pieces, C %% 32 == 0, W %% 4 == 0");
  const int K = 4 * C, HW = H * W;
  const int mtiles = cdiv(M, 128), kslices = K / 32, ptiles = cdiv(HW, 128);
  PRN_REQUIRE((int64_t)kslices * IMG_U4 * 16 < (1LL << 31) && (int64_t)C * HW < (1LL << 29) && (int64_t)B * M * HW * 4 < (1LL << 31), "prn_split_conv_up2: operand too large");
  if (int e = cut_weight(wp, images_ws, M, K, 4, (long long)M * K, PRN_PIECES_F16, st, "prn_split_conv_up2/prepare", 4)) return e;
  Split16Args a;
  a.img = (const uint4*)images_ws; a.ex = (const int*)((const char*)images_ws + (int64_t)4 * mtiles * kslices * IMG16_U4 * 16);
  a.x = x; a.bias = bias; a.addend = nullptr; a.y = y; a.partial = nullptr;
  a.M = M; a.K = K; a.B = B; a.HW = HW; a.epi = epi; a.mtiles = mtiles; a.kslices = kslices; a.ptiles = ptiles;
  a.total = mtiles * ptiles * B * 4; a.splits = 1; a.zx = 0; a.zy = 0; a.slice = 0;
  a.C = C; a.XH = H; a.XW = W; a.Wo = W; a.KW = 2; a.stride = 1; a.pad = 0;
  a.wide = 1; a.store_policy = 1;
  if (o->split_products >= 4) hipLaunchKernelGGL((split16_gemm_kernel<4, 2>), dim3(a.total, 1), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((split16_gemm_kernel<3, 2>), dim3(a.total, 1), dim3(256), 0, st, a);
  PRN_CHECK_LAUNCH("prn_split_conv_up2");
  return 0;
}

// Lab: weight gradient of a 1x1 convolution, dW[m][c] = sum_{b,p} dy[b][m][p] * x[b][c][p], on the bf16 matrix pipe with BOTH operands cut into
// three exact bf16 pieces inside the kernel (six products per multiply-add, fp32 accumulate) -- see planerecnet_amd/csrc/prn_gemm_split.hip for
// the arithmetic.  Both operands are pixel-contiguous, which is exactly the MFMA operand layout (a lane holds eight consecutive k of one row):
// tiles of 128 rows x 16 pixels go global -> LDS by buffer_load ... lds (chunk-swizzled by the SOURCE permutation), every wave reads its
// fragments as two ds_read_b128 per row block, cuts them in registers and feeds the pipe.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 wbf16x8_t;
typedef __attribute__((ext_vector_type(16))) float wf32x16_t;
typedef int wi32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void w_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
  const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
  const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
  const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
  h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
  m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
  l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}
__device__ __forceinline__ wi32x4_t w_make_desc(const void* p, unsigned bytes) {
  const unsigned long long q = (unsigned long long)p;
  wi32x4_t d;
  d.x = __builtin_amdgcn_readfirstlane((int)(unsigned)q);
  d.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(q >> 32) & 0xffff);
  d.z = __builtin_amdgcn_readfirstlane((int)bytes);
  d.w = 0x00020000;
  return d;
}
__device__ __forceinline__ void w_lds_dma16(unsigned lds_byte_addr, wi32x4_t desc, unsigned voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane((int)lds_byte_addr)), "v"(voff), "s"(desc),
               "s"(__builtin_amdgcn_readfirstlane(soff))
               : "memory");
}

struct WgSplitArgs {
  const float* x; const float* dy; float* part;      // part: [splits][M][C]
  int B, C, M, HW, tilesM, tilesC, splits, chunks;   // chunks = B * HW / 16
};

constexpr int WS_STAGE_BYTES = 16384;                // A tile (128 rows x 16 px x 4 B) + B tile
constexpr int WS_NST = 3;

template <int DBG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void wgrad_split_kernel(const WgSplitArgs a) {
  __shared__ uint4 lds[WS_NST * WS_STAGE_BYTES / 16];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tm = blockIdx.x % a.tilesM, tc = blockIdx.x / a.tilesM;
  const int sp = blockIdx.y;
  const int q0 = (int)((long long)a.chunks * sp / a.splits), q1 = (int)((long long)a.chunks * (sp + 1) / a.splits);
  const int HW = a.HW, cpi = HW >> 4;                 // chunks per image
  const wi32x4_t adesc = w_make_desc(a.dy, (unsigned)((long long)a.B * a.M * HW * 4));
  const wi32x4_t bdesc = w_make_desc(a.x, (unsigned)((long long)a.B * a.C * HW * 4));
  const unsigned lds0 = (unsigned)(unsigned long long)(void*)lds;
  // DMA assignment: instruction j of wave w covers LDS bytes [(j * 4 + w) * 1024, +1024) of a tile = rows 16 * (j * 4 + w) .. +15, lane -> (row = lane / 4,
  // slot = lane % 4); the slot holds source chunk slot ^ ((row >> 2) & 3)  (rows 4 apart share banks: the XOR spreads them)
  unsigned avoff[2], bvoff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rl = 16 * (j * 4 + wave) + (lane >> 2);
    const int ch = (lane & 3) ^ ((rl >> 2) & 3);
    const int m = tm * 128 + rl, c = tc * 128 + rl;
    avoff[j] = m < a.M ? (unsigned)(m * HW * 4 + ch * 16) : 0x80000000u;
    bvoff[j] = c < a.C ? (unsigned)(c * HW * 4 + ch * 16) : 0x80000000u;
  }
#define WS_DMA(q_, st_) do { \
    const int b_ = (q_) / cpi, p0_ = ((q_) - b_ * cpi) << 4; \
    const int sa_ = (b_ * a.M * HW + p0_) * 4, sb_ = (b_ * a.C * HW + p0_) * 4; \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) { \
      w_lds_dma16(lds0 + (unsigned)(st_) * WS_STAGE_BYTES + (unsigned)(j * 4 + wave) * 1024u, adesc, avoff[j], sa_); \
      w_lds_dma16(lds0 + (unsigned)(st_) * WS_STAGE_BYTES + 8192u + (unsigned)(j * 4 + wave) * 1024u, bdesc, bvoff[j], sb_); \
    } } while (0)
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, g = lane >> 5;
  wf32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int nq = q1 - q0;
  if (nq > 0) WS_DMA(q0, 0);
  if (nq > 1) WS_DMA(q0 + 1, 1);
  for (int i = 0; i < nq; ++i) {
    const int st = i % WS_NST;
    if (i + 1 < nq) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (i + 2 < nq) WS_DMA(q0 + i + 2, (i + 2) % WS_NST);
    const uint4* sa = lds + st * (WS_STAGE_BYTES / 16);
    const uint4* sb = sa + 512;
    wbf16x8_t ap[2][3], bp[2][3];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ra = wm * 64 + u * 32 + r, rb = wn * 64 + u * 32 + r;
      const int xa = (ra >> 2) & 3, xb = (rb >> 2) & 3;
      const uint4 a0 = sa[ra * 4 + ((2 * g) ^ xa)], a1 = sa[ra * 4 + ((2 * g + 1) ^ xa)];
      const uint4 b0 = sb[rb * 4 + ((2 * g) ^ xb)], b1 = sb[rb * 4 + ((2 * g + 1) ^ xb)];
      uint4 h, m, l;
      w_split2(__uint_as_float(a0.x), __uint_as_float(a0.y), h.x, m.x, l.x); w_split2(__uint_as_float(a0.z), __uint_as_float(a0.w), h.y, m.y, l.y);
      w_split2(__uint_as_float(a1.x), __uint_as_float(a1.y), h.z, m.z, l.z); w_split2(__uint_as_float(a1.z), __uint_as_float(a1.w), h.w, m.w, l.w);
      ap[u][0] = __builtin_bit_cast(wbf16x8_t, h); ap[u][1] = __builtin_bit_cast(wbf16x8_t, m); ap[u][2] = __builtin_bit_cast(wbf16x8_t, l);
      w_split2(__uint_as_float(b0.x), __uint_as_float(b0.y), h.x, m.x, l.x); w_split2(__uint_as_float(b0.z), __uint_as_float(b0.w), h.y, m.y, l.y);
      w_split2(__uint_as_float(b1.x), __uint_as_float(b1.y), h.z, m.z, l.z); w_split2(__uint_as_float(b1.z), __uint_as_float(b1.w), h.w, m.w, l.w);
      bp[u][0] = __builtin_bit_cast(wbf16x8_t, h); bp[u][1] = __builtin_bit_cast(wbf16x8_t, m); bp[u][2] = __builtin_bit_cast(wbf16x8_t, l);
    }
    if (DBG & 1) {
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i2][j][0] += __builtin_bit_cast(float, __builtin_bit_cast(uint4, ap[i2][0]).x ^ __builtin_bit_cast(uint4, bp[j][1]).y ^ __builtin_bit_cast(uint4, ap[i2][2]).z ^ __builtin_bit_cast(uint4, bp[j][2]).w ^ __builtin_bit_cast(uint4, ap[i2][1]).w);
      continue;
    }
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        wf32x16_t c = acc[i2][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i2][2], bp[j][0], c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i2][0], bp[j][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i2][1], bp[j][1], c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i2][1], bp[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i2][0], bp[j][1], c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i2][0], bp[j][0], c, 0, 0, 0);
        acc[i2][j] = c;
      }
  }
#undef WS_DMA
  float* pb = a.part + (long long)sp * a.M * a.C;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = tc * 128 + wn * 64 + j * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = tm * 128 + wm * 64 + i * 32 + (e >> 2) * 8 + g * 4 + (e & 3);
        if (row < a.M && col < a.C) pb[(long long)row * a.C + col] = acc[i][j][e];
      }
    }
}

__global__ void wgrad_reduce_kernel(const float* part, float* dw, long long n, int splits) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += part[(long long)k * n + i];
  dw[i] = s;
}

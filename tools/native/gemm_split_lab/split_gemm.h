// fp32 GEMM on the bf16 matrix pipe by exact three-way splitting (lab version 0).  See split_lab.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#ifndef SG_WPE
#define SG_WPE 2
#endif
#define SG_BM 128
#define SG_BN 128
#define SG_BK 32
#define SG_A_U4 1536          // uint4 per A image of one (m tile, k slice): 3 pieces x 4 k-groups x 128 rows x 16 B = 24 KB

// three exact bf16 pieces of two consecutive-k values, packed (low half = first value)
__device__ __forceinline__ void sg_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
  const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
  const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
  const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
  h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
  m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
  l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

// w[z][m][k] fp32 -> images [z][mtile][kslice][piece][kgroup][row][8 x bf16], zero padded in M and K
__global__ void sg_prepare_kernel(const float* __restrict__ w, uint4* __restrict__ ws, int M, int K, int mtiles, int kslices, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // one thread per (z, mtile, kslice, kgroup, row)
  if (i >= total) return;
  const int r = i % 128; const int g = (i / 128) % 4; const long long t = i / 512;
  const int ks = t % kslices; const long long zm = t / kslices; const int mt = zm % mtiles; const long long z = zm / mtiles;
  const int m = mt * 128 + r, k0 = ks * 32 + g * 8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (m < M && k0 + j < K) ? w[(z * M + m) * (long long)K + k0 + j] : 0.f;
  uint4 h, mm, l;
  sg_split2(v[0], v[1], h.x, mm.x, l.x); sg_split2(v[2], v[3], h.y, mm.y, l.y); sg_split2(v[4], v[5], h.z, mm.z, l.z); sg_split2(v[6], v[7], h.w, mm.w, l.w);
  uint4* o = ws + t * SG_A_U4;
  o[(0 * 4 + g) * 128 + r] = h; o[(1 * 4 + g) * 128 + r] = mm; o[(2 * 4 + g) * 128 + r] = l;
}

template <int NPROD, int DBG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SG_WPE, SG_WPE))) void sg_kernel(const uint4* __restrict__ wsA, const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ addend,
                                                    float* __restrict__ y, int M, int K, int B, int HW, int epi, int mtiles, int kslices, int ptiles) {
  __shared__ uint4 lds[2 * SG_A_U4];                                      // A image | B image (same layout, rows = pixels): 48 KB
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int pt = blockIdx.x % ptiles, b = blockIdx.x / ptiles;
  const int mt = blockIdx.y, z = blockIdx.z;
  const int p0 = pt * SG_BN;
  const float* xb = x + ((long long)z * B + b) * (long long)K * HW;
  const uint4* ag = wsA + ((long long)(z * mtiles + mt) * kslices) * SG_A_U4;
  const int g = wave, pp = lane;                                          // staging task: k-group (8 k) x pixel pair
  const int px = p0 + 2 * pp;
  const bool pvalid = px < HW;
  const int pxc = pvalid ? px : 0;
  float2 rb[8]; uint4 ra0, ra1, ra2, ra3, ra4, ra5;
#define SG_LOAD(ks_) do { \
    const uint4* ap_ = ag + (long long)(ks_) * SG_A_U4 + t; \
    ra0 = ap_[0]; ra1 = ap_[256]; ra2 = ap_[512]; ra3 = ap_[768]; ra4 = ap_[1024]; ra5 = ap_[1280]; \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) { \
      const int k = (ks_) * SG_BK + g * 8 + j; \
      rb[j] = *(const float2*)(xb + (long long)(k < K ? k : K - 1) * HW + pxc); \
    } } while (0)
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, gs = lane >> 5;
  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  SG_LOAD(0);
  for (int ks = 0; ks < kslices; ++ks) {
    __syncthreads();
    {
      lds[t] = ra0; lds[256 + t] = ra1; lds[512 + t] = ra2; lds[768 + t] = ra3; lds[1024 + t] = ra4; lds[1280 + t] = ra5;
      uint4 h0, m0, l0, h1, m1, l1;
#pragma unroll
      for (int j = 0; j < 8; ++j) if (!pvalid || ks * SG_BK + g * 8 + j >= K) rb[j] = make_float2(0.f, 0.f);
      if (DBG & 2) { h0 = make_uint4(__float_as_uint(rb[0].x), __float_as_uint(rb[1].x), __float_as_uint(rb[2].x), __float_as_uint(rb[3].x)); m0 = h0; l0 = h0; h1 = make_uint4(__float_as_uint(rb[4].y), __float_as_uint(rb[5].y), __float_as_uint(rb[6].y), __float_as_uint(rb[7].y)); m1 = h1; l1 = h1; } else {
      sg_split2(rb[0].x, rb[1].x, h0.x, m0.x, l0.x); sg_split2(rb[2].x, rb[3].x, h0.y, m0.y, l0.y); sg_split2(rb[4].x, rb[5].x, h0.z, m0.z, l0.z); sg_split2(rb[6].x, rb[7].x, h0.w, m0.w, l0.w);
      sg_split2(rb[0].y, rb[1].y, h1.x, m1.x, l1.x); sg_split2(rb[2].y, rb[3].y, h1.y, m1.y, l1.y); sg_split2(rb[4].y, rb[5].y, h1.z, m1.z, l1.z); sg_split2(rb[6].y, rb[7].y, h1.w, m1.w, l1.w); }
      uint4* o = lds + SG_A_U4 + g * 128 + 2 * pp;
      o[0 * 512] = h0; o[0 * 512 + 1] = h1; o[1 * 512] = m0; o[1 * 512 + 1] = m1; o[2 * 512] = l0; o[2 * 512 + 1] = l1;
    }
    __syncthreads();
    if (!(DBG & 4) && ks + 1 < kslices) SG_LOAD(ks + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int kg = 2 * s + gs;
      bf16x8_t a[2][3], bq[2][3];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint4 va = lds[(q * 4 + kg) * 128 + wm * 64 + i * 32 + r];
          const uint4 vb = lds[SG_A_U4 + (q * 4 + kg) * 128 + wn * 64 + i * 32 + r];
          a[i][q] = __builtin_bit_cast(bf16x8_t, va); bq[i][q] = __builtin_bit_cast(bf16x8_t, vb);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16_t c = acc[i][j];
          if (DBG & 1) { c[0] += __builtin_bit_cast(float, __builtin_bit_cast(uint4, a[i][0]).x ^ __builtin_bit_cast(uint4, bq[j][1]).y ^ __builtin_bit_cast(uint4, a[i][2]).z ^ __builtin_bit_cast(uint4, bq[j][2]).w ^ __builtin_bit_cast(uint4, a[i][1]).w ^ __builtin_bit_cast(uint4, bq[j][0]).x); acc[i][j] = c; continue; }
          if (NPROD >= 9) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], bq[j][2], c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], bq[j][2], c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], bq[j][1], c, 0, 0, 0); }
          if (NPROD >= 6) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], bq[j][0], c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bq[j][2], c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], bq[j][1], c, 0, 0, 0); }
          if (NPROD >= 3) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], bq[j][0], c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bq[j][1], c, 0, 0, 0); }
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bq[j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
  }
  float* yb = y + ((long long)z * B + b) * (long long)M * HW;
  const float* ab = addend ? addend + ((long long)z * B + b) * (long long)M * HW : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rbase = mt * SG_BM + wm * 64 + i * 32 + gs * 4;
    float bv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { const int row = rbase + (e >> 2) * 8 + (e & 3); bv[e] = bias ? bias[row < M ? row : M - 1] : 0.f; }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = p0 + wn * 64 + j * 32 + r;
      const bool cok = col < HW;
      const int colc = cok ? col : 0;
      float av[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) { const int row = rbase + (e >> 2) * 8 + (e & 3); av[e] = ab ? ab[(long long)(row < M ? row : M - 1) * HW + colc] : 0.f; }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = rbase + (e >> 2) * 8 + (e & 3);
        float v = acc[i][j][e] + bv[e] + av[e];
        if (epi == 1) v = fmaxf(v, 0.f);
        if (cok && row < M) yb[(long long)row * HW + col] = v;
      }
    }
  }
}


// ---- version 1: B operand straight from global memory into the MFMA layout (no LDS, no barrier for it), A images by LDS-DMA into a ring -------------
typedef int sg_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ sg_i32x4 sg_make_desc(const void* p, unsigned bytes) {
  const unsigned long long q = (unsigned long long)p;
  sg_i32x4 d;
  d.x = __builtin_amdgcn_readfirstlane((int)(unsigned)q);
  d.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(q >> 32) & 0xffff);
  d.z = __builtin_amdgcn_readfirstlane((int)bytes);
  d.w = 0x00020000;
  return d;
}
__device__ __forceinline__ void sg_lds_dma16(unsigned lds_byte_addr, sg_i32x4 desc, unsigned voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane((int)lds_byte_addr)), "v"(voff), "s"(desc),
               "s"(__builtin_amdgcn_readfirstlane(soff))
               : "memory");
}
#ifndef SG1_NST
#define SG1_NST 2
#endif
#ifndef SG1_WPE
#define SG1_WPE 3
#endif
template <int NPROD, int DBG, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 2 : SG1_WPE, NW == 8 ? 2 : SG1_WPE))) void sg1_kernel(const uint4* __restrict__ wsA, const float* __restrict__ x, const float* __restrict__ bias,
                                                                                                   const float* __restrict__ addend, float* __restrict__ y, int M, int K, int B, int HW, int epi,
                                                                                                   int mtiles, int kslices, int ptiles, int total) {
  __shared__ uint4 lds[SG1_NST * SG_A_U4];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  int id = blockIdx.x;
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);              // the eight m-tiles / neighbours of one XCD's L2 stay together
  const int mt = id % mtiles; const int rest = id / mtiles;
  const int pt = rest % ptiles; const int zb = rest / ptiles; const int b = zb % B, z = zb / B;
  const int p0 = pt * (NW * 32) + wave * 32;
  const int r = lane & 31, gs = lane >> 5;
  const int px = p0 + r;
  const int pxc = px < HW ? px : HW - 1;
  const float* xb = x + ((long long)z * B + b) * (long long)K * HW;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, K * HW * 4, 0x00020000);
  const int xoff = (pxc + gs * 8 * HW) * 4;                                    // this lane's byte offset inside a 16-row group
  const uint4* ag = wsA + ((long long)(z * mtiles + mt) * kslices) * SG_A_U4;
  const sg_i32x4 adesc = sg_make_desc(ag, (unsigned)kslices * SG_A_U4 * 16u);
  const unsigned lds0 = (unsigned)(unsigned long long)(void*)lds;             // LDS byte address of the ring
  float rn[16];
  bf16x8_t bp[2][3];
#define SG1_DMA(ks_, st_) do { \
    _Pragma("unroll") for (int i = 0; i < 24 / NW; ++i) \
      sg_lds_dma16(lds0 + (unsigned)(st_) * (SG_A_U4 * 16) + (unsigned)(i * NW + wave) * 1024u, adesc, (unsigned)lane * 16u, ((ks_) * SG_A_U4 + (i * NW + wave) * 64) * 16); \
  } while (0)
#define SG1_LOADB(dst_, ks_) do { \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) { \
        dst_[s2 * 8 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xoff, ((ks_) * SG_BK + s2 * 16 + j) * HW * 4, 0)); \
      } } while (0)
#define SG1_SPLIT(src_) do { \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) { \
      uint4 h, m, l; \
      sg_split2(src_[s2 * 8 + 0], src_[s2 * 8 + 1], h.x, m.x, l.x); sg_split2(src_[s2 * 8 + 2], src_[s2 * 8 + 3], h.y, m.y, l.y); \
      sg_split2(src_[s2 * 8 + 4], src_[s2 * 8 + 5], h.z, m.z, l.z); sg_split2(src_[s2 * 8 + 6], src_[s2 * 8 + 7], h.w, m.w, l.w); \
      bp[s2][0] = __builtin_bit_cast(bf16x8_t, h); bp[s2][1] = __builtin_bit_cast(bf16x8_t, m); bp[s2][2] = __builtin_bit_cast(bf16x8_t, l); \
    } } while (0)
#define SG1_STEP(s2_) do { \
    const int kg = 2 * (s2_) + gs; \
    const bf16x8_t bh = bp[s2_][0], bm = bp[s2_][1], bl = bp[s2_][2]; \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { \
      const uint4* ap = lds + st * SG_A_U4 + kg * 128 + i * 32 + r; \
      const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, ap[0]), am = __builtin_bit_cast(bf16x8_t, ap[512]), al = __builtin_bit_cast(bf16x8_t, ap[1024]); \
      f32x16_t c = acc[i]; \
      if (DBG & 1) { c[0] += __builtin_bit_cast(float, __builtin_bit_cast(uint4, ah).x ^ __builtin_bit_cast(uint4, am).y ^ __builtin_bit_cast(uint4, al).z ^ __builtin_bit_cast(uint4, bh).x ^ __builtin_bit_cast(uint4, bm).y ^ __builtin_bit_cast(uint4, bl).z); acc[i] = c; continue; } \
      if (NPROD >= 9) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bl, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bm, c, 0, 0, 0); } \
      if (NPROD >= 6) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0); } \
      if (NPROD >= 3) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0); } \
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0); \
      acc[i] = c; \
    } } while (0)
  f32x16_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  SG1_DMA(0, 0);
  SG1_LOADB(rn, 0);
  SG1_SPLIT(rn);
  for (int ks = 0; ks < kslices; ++ks) {
    const int st = ks % SG1_NST;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool more = ks + 1 < kslices;
    const int kn = more ? ks + 1 : ks;
    if (more) SG1_DMA(ks + 1, (ks + 1) % SG1_NST);
    if (!(DBG & 4)) SG1_LOADB(rn, kn);
    __builtin_amdgcn_sched_barrier(0);
    SG1_STEP(0);
    __builtin_amdgcn_sched_barrier(0);
    SG1_STEP(1);
    SG1_SPLIT(rn);
  }
  if (DBG & 8) { float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) sum += acc[i][e];
    if (sum == 123.456f) y[t] = sum;
    return; }
  float* yb = y + ((long long)z * B + b) * (long long)M * HW;
  const float* ab = addend ? addend + ((long long)z * B + b) * (long long)M * HW : nullptr;
  const bool cok = px < HW;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rbase = mt * SG_BM + i * 32 + gs * 4;
    float bv[16], av[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { const int row = rbase + (e >> 2) * 8 + (e & 3); const int rc = row < M ? row : M - 1; bv[e] = bias ? bias[rc] : 0.f; av[e] = ab ? ab[(long long)rc * HW + pxc] : 0.f; }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = rbase + (e >> 2) * 8 + (e & 3);
      float v = acc[i][e] + bv[e] + av[e];
      if (epi == 1) v = fmaxf(v, 0.f);
      if (cok && row < M) yb[(long long)row * HW + px] = v;
    }
  }
}

static inline long long split_gemm_ws_bytes(int M, int K, int Z) { return (long long)Z * ((M + 127) / 128) * ((K + 31) / 32) * SG_A_U4 * 16; }
static inline void split_gemm_prepare(const float* w, void* ws, int M, int K, int Z, hipStream_t st) {
  const int mtiles = (M + 127) / 128, kslices = (K + 31) / 32;
  const long long total = (long long)Z * mtiles * kslices * 512;
  sg_prepare_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(w, (uint4*)ws, M, K, mtiles, kslices, total);
}
static inline void split_gemm_run(const void* ws, const float* x, const float* bias, const float* addend, float* y, int M, int K, int B, int HW, int Z, int epi, int nprod, hipStream_t st) {
  const int mtiles = (M + 127) / 128, kslices = (K + 31) / 32, ptiles = (HW + SG_BN - 1) / SG_BN;
  dim3 grid(ptiles * B, mtiles, Z);
  static const int dbg = getenv("SG_DBG") ? atoi(getenv("SG_DBG")) : 0;
  static const int ver = getenv("SG_V") ? atoi(getenv("SG_V")) : 1;
  if (ver == 1 || ver == 8) {
    const int nw = ver == 8 ? 8 : 4;
    const int pt1 = (HW + nw * 32 - 1) / (nw * 32);
    const int total = pt1 * B * mtiles * Z;
#define SG1_GO(NP, D) do { if (nw == 8) sg1_kernel<NP, D, 8><<<total, 512, 0, st>>>((const uint4*)ws, x, bias, addend, y, M, K, B, HW, epi, mtiles, kslices, pt1, total); \
                           else sg1_kernel<NP, D, 4><<<total, 256, 0, st>>>((const uint4*)ws, x, bias, addend, y, M, K, B, HW, epi, mtiles, kslices, pt1, total); } while (0)
    if (dbg == 1) SG1_GO(6, 1); else if (dbg == 4) SG1_GO(6, 4); else if (dbg == 5) SG1_GO(6, 5); else if (dbg == 8) SG1_GO(6, 8); else if (dbg == 13) SG1_GO(6, 13);
    else if (nprod >= 9) SG1_GO(9, 0); else if (nprod >= 6) SG1_GO(6, 0); else SG1_GO(3, 0);
    return;
  }
#define SG_GO(NP, D) sg_kernel<NP, D><<<grid, 256, 0, st>>>((const uint4*)ws, x, bias, addend, y, M, K, B, HW, epi, mtiles, kslices, ptiles)
  if (dbg == 1) SG_GO(6, 1); else if (dbg == 2) SG_GO(6, 2); else if (dbg == 4) SG_GO(6, 4); else if (dbg == 3) SG_GO(6, 3); else if (dbg == 6) SG_GO(6, 6); else if (dbg == 7) SG_GO(6, 7);
  else if (nprod >= 9) SG_GO(9, 0); else if (nprod >= 6) SG_GO(6, 0); else if (nprod >= 3) SG_GO(3, 0); else SG_GO(1, 0);
}

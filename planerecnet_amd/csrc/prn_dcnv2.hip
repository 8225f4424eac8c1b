// DCNv2 (modulated deformable 3x3 convolution) as ONE operator on gfx950: the bilinear 4-corner gather of
// torchvision.ops.deform_conv2d is the operand loader of the MFMA contraction -- no column tensor exists in HBM.
//
//   table   : one small launch turns (offset, mask) into a per-(output pixel, tap) gather table: four byte offsets into
//             the image's channel-0 plane (0x80000000 = corner outside the image -> the buffer range check returns 0.0
//             without a memory access) and four weights (mask * bilinear weight).  32 B per entry, 9 entries per pixel.
//   forward : Y[M x N] = W[M x 9C] * cols[9C x N] with cols[(c,t), n] = sum_q wt[t,n,q] * x[c, off[t,n,q]] evaluated while
//             the K slice is staged into LDS: the workgroup keeps the table rows of its 64 pixels in LDS (18 KB), the channel
//             offset is a scalar (soffset), so an operand element costs one ds_read_b128 (offsets), four buffer_load_dword,
//             one ds_read_b128 (weights) and four FMAs -- VALU / TA work that runs beside the fp32 MFMA pipe (64 cycles
//             per v_mfma_f32_32x32x2_f32, i.e. 32 VALU issue slots).
//   d-weight: dW[M x 9C] = dY[M x N] * cols^T: the same gather in the weight-gradient GEMM's operand loader; the table
//             rows of a 16-pixel chunk (4.6 KB) are staged two chunks ahead, the gathers one chunk ahead of the MFMA loop.
//   d-input / d-offset / d-mask: the column GRADIENT  W^T dY  [9C x N] is still materialised once (transient workspace):
//             evaluating it twice (once per consumer) would cost a second 2*M*9C*N GEMM on a 157 TFLOP/s fp32 pipe, i.e.
//             ~110 us for 256 ch @ 30x40, against ~50 us for writing + re-reading the 88 MB tensor.
#include <stdlib.h>
#include <type_traits>
#include "prn_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 bload2(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  const f32x2 q = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, soff, 0));
  return make_float2(q.x, q.y);
}
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
  return make_float4(q.x, q.y, q.z, q.w);
}

// ------------------------------------------------------------------------------------------------ gather table
// Layout (float4 units): tab[((chunk * 9 + t) * 2 + s) * 16 + p], chunk = n / 16, p = n % 16, s = 0: byte offsets of the two
// row pairs (as uint bit patterns in .x / .y; .z / .w unused), s = 1: the four slot weights (row 0 left / right, row 1 left / right).  One 16-pixel chunk is 4608 contiguous bytes (what the weight-gradient
// kernel stages per step); the forward kernel's 64-pixel tile is four consecutive chunks.  Pixels up to the next multiple
// of 64 exist in the table and read as "all corners outside".
constexpr int TAB_CHUNK4 = 9 * 2 * 16;       // float4 per chunk

struct TabArgs {
  const float* off; const float* msk; float4* tab;
  int B, C, H, W, Ho, Wo, stride, pad, raw;
  int64_t off_bs, msk_bs;     // elements per image in offset / mask (raw: both 27 * Ho * Wo)
  float maxoff;
  int N, Npad;
};

__device__ __forceinline__ void dcnv2_table_body(const TabArgs& a, int block) {
  const int gid = block * 256 + threadIdx.x;
  if (gid >= a.Npad * 9) return;
  const int p = gid & 15, t = (gid >> 4) % 9, chunk = gid / 144;
  const int n = chunk * 16 + p;
  uint4 o = make_uint4(OOB, OOB, 0u, 0u);
  float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < a.N) {
    const int plane = a.Ho * a.Wo;
    const int b = n / plane, pix = n - b * plane, ho = pix / a.Wo, wo = pix - ho * a.Wo;
    const float* ob = a.off + (size_t)b * a.off_bs;
    float dy = ob[(size_t)(2 * t) * plane + pix], dx = ob[(size_t)(2 * t + 1) * plane + pix];
    float mod = 1.f;
    if (a.raw) {                                           // models/dcn.py:53-57: clamp(+-max_offset), 2 * sigmoid
      dy = fminf(fmaxf(dy, -a.maxoff), a.maxoff);
      dx = fminf(fmaxf(dx, -a.maxoff), a.maxoff);
      mod = 2.f / (1.f + expf(-ob[(size_t)(18 + t) * plane + pix]));
    } else if (a.msk) {
      mod = a.msk[(size_t)b * a.msk_bs + (size_t)t * plane + pix];
    }
    const int ki = t / 3, kj = t - ki * 3;
    const float y = (float)(ho * a.stride - a.pad + ki) + dy, x = (float)(wo * a.stride - a.pad + kj) + dx;
    const bool inside = (y > -1.f) && (y < (float)a.H) && (x > -1.f) && (x < (float)a.W);
    const float fy = floorf(y), fx = floorf(x);
    const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
    const float ly = y - fy, lx = x - fx, hy = 1.f - ly, hx = 1.f - lx;
    const bool vy0 = inside && y0 >= 0, vy1 = inside && y1 <= a.H - 1, vx0 = x0 >= 0, vx1 = x1 <= a.W - 1;
    // The two corners of a row are neighbours in memory: ONE 8-byte load per row fetches the pixel pair (xb, xb + 1) with
    // xb = clamp(x0, 0, W - 2), always inside the row.  The weights are stored per loaded SLOT: at the left / right image
    // border the valid corner moves to the other slot and the slot that holds a pixel the sample does not use gets 0.
    const int xb = x0 < 0 ? 0 : (x0 > a.W - 2 ? (a.W - 2 > 0 ? a.W - 2 : 0) : x0);
    const float wl = (vx0 && x0 == xb) ? hx : ((vx1 && x1 == xb) ? lx : 0.f);            // weight of pixel xb
    const float wr = (vx1 && x1 == xb + 1) ? lx : ((vx0 && x0 == xb + 1) ? hx : 0.f);    // weight of pixel xb + 1
    const unsigned base = (unsigned)b * (unsigned)a.C * (unsigned)(a.H * a.W);
    if (vy0) { o.x = (base + (unsigned)(y0 * a.W + xb)) * 4u; w.x = mod * hy * wl; w.y = mod * hy * wr; }
    if (vy1) { o.y = (base + (unsigned)(y1 * a.W + xb)) * 4u; w.z = mod * ly * wl; w.w = mod * ly * wr; }
  }
  const size_t e = ((size_t)(chunk * 9 + t) * 2) * 16 + p;
  a.tab[e] = make_float4(__uint_as_float(o.x), __uint_as_float(o.y), __uint_as_float(o.z), __uint_as_float(o.w));
  a.tab[e + 16] = w;
}

// ------------------------------------------------------------------------------------------------ forward
struct FwdArgs {
  const float* x; const float* w; const float* bias; const float4* tab; float* y; float* ws;
  int B, C, HW, M, K, N, HoWo;
  int tilesM, nblocks, splits, epi;
  int xbytes, wbytes;
};

// Tile (64 * TM) x 64 output channels x pixels, 256 threads = 2 x 2 waves, each wave TM/… see below; K slices of 16, LDS
// double buffered, next slice's loads in flight during the MFMA loop (same schedule as conv_igemm_kernel in prn_conv.hip).
// WGM = waves along M: 2 (wave tile 32*TM x 32) for TM <= 2; TM = 4 keeps the 2 x 2 grid as well (wave tile 128 x 32).
template <int TM, bool K4, int BK = 16>
__global__ __launch_bounds__(256, (TM == 1 ? 4 : 3)) void dcnv2_fwd_kernel(FwdArgs a) {
  constexpr int BM = 64 * TM, BN = 64, LDA = BK + 1;
  constexpr int NB = BK / 4;       // gathered operand elements per thread per K slice: K rows krow0 + 4 * i
  constexpr int AQ = BK / 4;       // float4 groups per weight row
  constexpr int AROWS = 256 / AQ;  // weight rows covered by one sweep of the workgroup
  constexpr int NA = BM / AROWS;   // float4 weight groups per thread per K slice: rows arow + AROWS * i
  __shared__ float As[2][BM * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];
  __shared__ uint2 toff[9 * BN];
  __shared__ float4 twt[9 * BN];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int id = prn_xcd_remap(blockIdx.x, a.nblocks);
  const int m0 = (id % a.tilesM) * BM, n0 = (id / a.tilesM) * BN;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.x, a.xbytes), wr = make_rsrc(a.w, a.wbytes);

  {  // table rows of this tile's 64 pixels: four chunks, contiguous in HBM; LDS layout [tap][pixel] (16-byte lanes: the
     // ds_read_b128 of 16 consecutive lanes covers all 64 banks)
    const float4* tg = a.tab + (size_t)(n0 >> 4) * TAB_CHUNK4;
    for (int i = tid; i < 4 * TAB_CHUNK4; i += 256) {
      const int cl = i / TAB_CHUNK4, r = i - cl * TAB_CHUNK4, t = r >> 5, s = (r >> 4) & 1, p = r & 15;
      const float4 v = tg[i];
      if (s) twt[t * BN + cl * 16 + p] = v;
      else toff[t * BN + cl * 16 + p] = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));
    }
    __syncthreads();
  }

  const int nl = tid & 63;
  const int krow0 = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform K row inside a sweep
  const int arow = tid / AQ, akq = (tid % AQ) * 4;
  unsigned abase[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int m = m0 + arow + AROWS * i;
    abase[i] = (m < a.M) ? (unsigned)(m * a.K + akq) * 4u : OOB;
  }

  float ra[NA][4];
  float g[NB][4];
  f32x16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  auto load_tile = [&](int k0) {
    const int k = k0 + akq;
    if (K4) {
      const unsigned tail = k < a.K ? 0u : OOB;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const float4 v = bload4(wr, abase[i] | tail, k0 * 4);
        ra[i][0] = v.x; ra[i][1] = v.y; ra[i][2] = v.z; ra[i][3] = v.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[i][j] = bload(wr, (k + j < a.K) ? abase[i] + 4u * j : OOB, k0 * 4);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int kr = k0 + krow0 + 4 * i;                   // wave-uniform: channel and tap are scalars
      const int c = kr / 9, t = kr - c * 9;
      const uint2 o = toff[t * BN + nl];
      const unsigned dead = kr < a.K ? 0u : OOB;
      const int so = c * a.HW * 4;
      const float2 r0 = bload2(xr, o.x | dead, so), r1 = bload2(xr, o.y | dead, so);
      g[i][0] = r0.x; g[i][1] = r0.y; g[i][2] = r1.x; g[i][3] = r1.y;
    }
  };
  auto store_tile = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) As[buf][(arow + AROWS * i) * LDA + akq + j] = ra[i][j];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int kr = k0 + krow0 + 4 * i;
      const int c = kr / 9, t = kr - c * 9;
      const float4 wq = twt[t * BN + nl];
      Bs[buf][(krow0 + 4 * i) * BN + nl] = (wq.x * g[i][0] + wq.y * g[i][1]) + (wq.z * g[i][2] + wq.w * g[i][3]);
    }
  };
  auto mma_tile = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float av[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = As[buf][(wm * TM * 32 + i * 32 + (lane & 31)) * LDA + kk * 2 + (lane >> 5)];
      const float bv = Bs[buf][(kk * 2 + (lane >> 5)) * BN + wn * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv, acc[i], 0, 0, 0);
    }
  };
  const int KTall = (a.K + BK - 1) / BK;
  const int kt0 = (int)((int64_t)blockIdx.y * KTall / a.splits), KT = (int)((int64_t)(blockIdx.y + 1) * KTall / a.splits);
  load_tile(kt0 * BK);
  store_tile(kt0 & 1, kt0 * BK);
  __syncthreads();
  for (int kt = kt0; kt + 1 < KT; ++kt) {
    load_tile((kt + 1) * BK);
    __builtin_amdgcn_sched_barrier(0);
    mma_tile(kt & 1);
    __builtin_amdgcn_sched_barrier(0);
    store_tile((kt + 1) & 1, (kt + 1) * BK);
    __syncthreads();
  }
  mma_tile((KT - 1) & 1);

  // epilogue (C/D layout: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5))
  const bool partial = a.splits > 1;
  float* __restrict__ outp = partial ? a.ws + (size_t)blockIdx.y * a.B * a.M * a.HoWo : a.y;
  const bool has_bias = !partial && a.bias != nullptr;
  const int epi = partial ? PRN_EPI_NONE : a.epi;
  const int nn = n0 + wn * 32 + (lane & 31);
  if (nn < a.N) {
    const int bb = nn / a.HoWo, p = nn - bb * a.HoWo;
    const size_t base = (size_t)bb * a.M * a.HoWo + p;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mbase = m0 + wm * TM * 32 + i * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2);
        if (m >= a.M) continue;
        float v = acc[i][r];
        if (has_bias) v += a.bias[m];
        if (epi == PRN_EPI_RELU) v = fmaxf(v, 0.f);
        outp[base + (size_t)m * a.HoWo] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct WgArgs {
  const float* x; const float* dy; const float4* tab; float* out;
  int B, C, HW, M, K, N, HoWo;
  int tilesM, tilesJ, splits, chunks;
  int xbytes, dybytes, tabbytes;
  int ilv;                    // reduction loop with the staging work issued inside the MFMA loop
};

// dW tile (64 * TM) x 64 (output channels x (c, tap) columns), reduction over 16-pixel chunks; thread (nl = tid & 15,
// jrow = tid >> 4) samples columns jrow + 16 * i of pixel nl.  Three-stage software pipeline per chunk ch:
//   table rows of chunk ch + 2: HBM -> registers (start of the step) -> LDS (end of the step)
//   corner gathers + dY rows of chunk ch + 1: issued at the start of the step from the LDS table written one step earlier,
//                                             interpolated and stored to LDS after the MFMA loop
//   MFMA loop on chunk ch.
template <int TM, bool N4>
__global__ __launch_bounds__(256, (TM == 1 ? 4 : (TM == 2 ? 3 : 2))) void dcnv2_wgrad_kernel(WgArgs a) {
  constexpr int BM = 64 * TM, BJ = 64, LD = 17, NBJ = 4;
  __shared__ float As[2][BM * LD];
  __shared__ float Bs[2][BJ * LD];
  __shared__ uint2 toff[3][9 * 16];                 // (two buffers in the phase-separated loop, a ring of three in the interleaved one)
  __shared__ float4 twt[3][9 * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wj = wave & 1;
  const int m0 = (blockIdx.x % a.tilesM) * BM, j0 = (blockIdx.x / a.tilesM) * BJ;
  const int split = blockIdx.y;
  const int cbeg = (int)((int64_t)split * a.chunks / a.splits), cend = (int)((int64_t)(split + 1) * a.chunks / a.splits);
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.x, a.xbytes), dyr = make_rsrc(a.dy, a.dybytes), tr = make_rsrc(a.tab, a.tabbytes);
  const int arow = tid >> 2, anq = (tid & 3) * 4;
  const int nl = tid & 15, jrow = tid >> 4;
  unsigned jcoff[NBJ];              // channel byte offset (or OOB for columns beyond K)
  int jt[NBJ];                      // tap
#pragma unroll
  for (int i = 0; i < NBJ; ++i) {
    const int j = j0 + jrow + 16 * i;
    const int c = j / 9;
    jt[i] = j - c * 9;
    jcoff[i] = j < a.K ? (unsigned)(c * a.HW) * 4u : OOB;
  }
  bool mok[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) mok[i] = m0 + arow + 64 * i < a.M;

  float ra[TM][4], rb[NBJ][4];
  float4 rt0, rt1;
  f32x16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  int a_b, a_p;                     // dY cursor: image and pixel of this thread's 4-pixel group
  {
    const int n = cbeg * 16 + anq;
    a_b = n / a.HoWo; a_p = n - a_b * a.HoWo;
  }
  auto load_table = [&](int ch) {                          // -> registers; chunks past the range read as zeros (never used)
    const unsigned ok = ch < cend ? 0u : OOB;
    rt0 = bload4(tr, ((unsigned)tid * 16u) | ok, ch * (TAB_CHUNK4 * 16));
    rt1 = bload4(tr, (tid < TAB_CHUNK4 - 256 ? (unsigned)(tid + 256) * 16u : OOB) | ok, ch * (TAB_CHUNK4 * 16));
  };
  auto write_table = [&](int tb) {
    auto put = [&](int r, float4 v) {
      const int t = r >> 5, s = (r >> 4) & 1, p = r & 15;
      if (s) twt[tb][t * 16 + p] = v;
      else toff[tb][t * 16 + p] = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));
    };
    put(tid, rt0);
    if (tid < TAB_CHUNK4 - 256) put(tid + 256, rt1);
  };
  auto load_chunk = [&](int ch, int tb) {
    {  // dY rows: 4 consecutive pixels of one output channel
      const int n = ch * 16 + anq;
      if (N4) {
        const bool ok = n < a.N;
        const unsigned base = (unsigned)((a_b * a.M + m0 + arow) * a.HoWo + a_p) * 4u;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float4 v = bload4(dyr, (ok && mok[i]) ? base : OOB, i * 64 * a.HoWo * 4);
          ra[i][0] = v.x; ra[i][1] = v.y; ra[i][2] = v.z; ra[i][3] = v.w;
        }
      } else {
        int qb = a_b, qp = a_p;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = n + q < a.N;
          const unsigned base = (unsigned)((qb * a.M + m0 + arow) * a.HoWo + qp) * 4u;
#pragma unroll
          for (int i = 0; i < TM; ++i) ra[i][q] = bload(dyr, (ok && mok[i]) ? base : OOB, i * 64 * a.HoWo * 4);
          if (++qp >= a.HoWo) { qp = 0; ++qb; }
        }
      }
      a_p += 16;
      while (a_p >= a.HoWo) { a_p -= a.HoWo; ++a_b; }
    }
#pragma unroll
    for (int i = 0; i < NBJ; ++i) {                        // sampled columns: four corners each
      const uint2 o = toff[tb][jt[i] * 16 + nl];
      const float2 r0 = bload2(xr, o.x + jcoff[i], 0), r1 = bload2(xr, o.y + jcoff[i], 0);   // (OOB + channel offset stays >= 2^31)
      rb[i][0] = r0.x; rb[i][1] = r0.y; rb[i][2] = r1.x; rb[i][3] = r1.y;
    }
  };
  auto store_chunk = [&](int buf, int tb) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) As[buf][(arow + 64 * i) * LD + anq + q] = ra[i][q];
#pragma unroll
    for (int i = 0; i < NBJ; ++i) {
      const float4 wq = twt[tb][jt[i] * 16 + nl];
      Bs[buf][(jrow + 16 * i) * LD + nl] = (wq.x * rb[i][0] + wq.y * rb[i][1]) + (wq.z * rb[i][2] + wq.w * rb[i][3]);
    }
  };
  auto mma_chunk = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      float av[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = As[buf][(wm * TM * 32 + i * 32 + (lane & 31)) * LD + kk * 2 + (lane >> 5)];
      const float bv = Bs[buf][(wj * 32 + (lane & 31)) * LD + kk * 2 + (lane >> 5)];
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv, acc[i], 0, 0, 0);
    }
  };

  if (N4 && a.ilv && cend - cbeg >= 2) {
    // Interleaved schedule: under every k step of chunk ch's MFMAs one operand piece of chunk ch + 1 goes from registers to LDS (a dY float4
    // group as it is; a sampled column after its four corners are weighted with the table of chunk ch + 1) and is re-fetched for chunk ch + 2 (the
    // gather offsets come from the table of chunk ch + 2).  Tables therefore live in a ring of three LDS buffers: chunk c's table is written in iteration
    // c - 3 (from registers loaded an iteration earlier), its offsets are read in iteration c - 2, its weights in iteration c - 1.
    constexpr int NP = TM + NBJ;
    load_table(cbeg); write_table(0);
    load_table(cbeg + 1); write_table(1);
    load_table(cbeg + 2); write_table(2);
    __syncthreads();
    load_chunk(cbeg, 0);
    store_chunk(0, 0);
    load_chunk(cbeg + 1, 1);
    load_table(cbeg + 3);
    __syncthreads();
    for (int ch = cbeg; ch + 1 < cend; ++ch) {
      const int it = ch - cbeg, buf = it & 1, nbuf = buf ^ 1;
      const int t1 = (it + 1) % 3, t2 = (it + 2) % 3;        // table buffers of chunks ch + 1 (weights) and ch + 2 (offsets)
      const bool aok = (ch + 2) * 16 + anq < a.N;
      const unsigned abase2 = (unsigned)((a_b * a.M + m0 + arow) * a.HoWo + a_p) * 4u;
      float av[2][TM], bv[2];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[0][i] = As[buf][(wm * TM * 32 + i * 32 + (lane & 31)) * LD + (lane >> 5)];
      bv[0] = Bs[buf][(wj * 32 + (lane & 31)) * LD + (lane >> 5)];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        if (kk + 1 < 8) {
#pragma unroll
          for (int i = 0; i < TM; ++i) av[(kk + 1) & 1][i] = As[buf][(wm * TM * 32 + i * 32 + (lane & 31)) * LD + (kk + 1) * 2 + (lane >> 5)];
          bv[(kk + 1) & 1] = Bs[buf][(wj * 32 + (lane & 31)) * LD + (kk + 1) * 2 + (lane >> 5)];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1][i], bv[kk & 1], acc[i], 0, 0, 0);
#pragma unroll
        for (int p = 0; p < NP; ++p)
          if ((p * 8) / NP == kk) {
            if (p < TM) {
              const int i = p < TM ? p : 0;
#pragma unroll
              for (int q = 0; q < 4; ++q) As[nbuf][(arow + 64 * i) * LD + anq + q] = ra[i][q];
              const float4 v = bload4(dyr, (aok && mok[i]) ? abase2 : OOB, i * 64 * a.HoWo * 4);
              ra[i][0] = v.x; ra[i][1] = v.y; ra[i][2] = v.z; ra[i][3] = v.w;
            } else {
              const int i = p >= TM ? p - TM : 0;
              const float4 wq = twt[t1][jt[i] * 16 + nl];
              Bs[nbuf][(jrow + 16 * i) * LD + nl] = (wq.x * rb[i][0] + wq.y * rb[i][1]) + (wq.z * rb[i][2] + wq.w * rb[i][3]);
              const uint2 o = toff[t2][jt[i] * 16 + nl];
              const float2 r0 = bload2(xr, o.x + jcoff[i], 0), r1 = bload2(xr, o.y + jcoff[i], 0);
              rb[i][0] = r0.x; rb[i][1] = r0.y; rb[i][2] = r1.x; rb[i][3] = r1.y;
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      a_p += 16;
      while (a_p >= a.HoWo) { a_p -= a.HoWo; ++a_b; }
      write_table(it % 3);                                   // table of chunk ch + 3 (its buffer held chunk ch's: done with since the previous iteration)
      load_table(ch + 4);
      __syncthreads();
    }
    mma_chunk((cend - 1 - cbeg) & 1);
  } else if (cbeg < cend) {
    load_table(cbeg);
    write_table(0);
    load_table(cbeg + 1);
    write_table(1);
    __syncthreads();
    load_chunk(cbeg, 0);
    store_chunk(0, 0);
    __syncthreads();
    for (int ch = cbeg; ch + 1 < cend; ++ch) {
      const int q = (ch - cbeg) & 1;                        // LDS buffers of chunk ch: As/Bs[q], table[q]
      load_chunk(ch + 1, q ^ 1);
      load_table(ch + 2);
      __builtin_amdgcn_sched_barrier(0);
      mma_chunk(q);
      __builtin_amdgcn_sched_barrier(0);
      store_chunk(q ^ 1, q ^ 1);
      write_table(q);                                      // chunk ch's table rows were last read by the previous step's store
      __syncthreads();
    }
    mma_chunk((cend - 1 - cbeg) & 1);
  }

  float* out = a.out + (size_t)split * a.M * a.K;
  const int jj = j0 + wj * 32 + (lane & 31);
  if (jj < a.K) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < a.M) out[(size_t)m * a.K + jj] = acc[i][r];
      }
  }
}

// ------------------------------------------------------------------------------------------------ windowed forward (round 5)
// The gathers above fetch every operand element from L2 (two 8-byte loads for ONE sampled value, 16 KB of gathered bytes per 8-deep K slice
// of a 256 x 64 tile) one slice ahead of the MFMA loop: the launch is bound by that round trip, not by the matrix pipe (the same operand on
// the 16-bit pipe ran no faster: DESIGN 9.6).  Here the pixel tile is a PATCH (PH x PW output pixels of one image, PH * PW = 64) and the
// input WINDOW the patch's 9 x 64 sampling points fall into -- (PH - 1) * stride + 3 rows plus the offsets' spread -- is staged in LDS per
// channel: a K slice is TWO channels x 9 taps = 18 rows, its windows are fetched once (coalesced row segments, ~0.1-0.4 KB per channel
// instead of 4.6 KB of gathers), two slices ahead, and the bilinear gather is two ds_read2_b32 per element.  A thread samples the same
// (tap, pixel) pairs in every slice, so its table entries (LDS offset of the 2 x 2 footprint, four weights) live in registers for the
// whole K loop.  The window is zero-padded where it leaves the image, which IS torchvision's rule (corners outside contribute nothing; a
// point at or beyond -1 / H / W has all its weight on zero rows) -- no validity select in the loop.
//   table2 : one workgroup per patch: sampling geometry of its 9 x 64 points, bounding box of the live ones (LDS min / max) -> window
//            origin and extent; entries {LDS offset | corner validity, global byte offset of the top-left corner} + 4 weights.
//   fallback: a patch whose window exceeds WROWS x WPITCH (offsets are clamped at +-max(h, w) / 4, so it can) samples from global memory
//            with four guarded dword loads per element -- same weights, same summation order, bit-identical to the window path.
constexpr int WPITCH = 40;                   // floats per window row in LDS: == 8 (mod 32), the 4 x 8 lanes of a half-wave hit 32 banks
constexpr int WROWS = 32;
constexpr int WCH = WPITCH * WROWS;          // floats per channel window
constexpr int W2_BK = 18;                    // two channels x nine taps
constexpr int TILE2_F4 = 2 + 2 * 9 * 64;     // float4 per patch in the table: header (8 ints), meta[9][64], weights[9][64]

struct Tab2Args {
  const float* off; const float* msk; float4* tab;
  int B, C, H, W, Ho, Wo, stride, pad, raw;
  int64_t off_bs, msk_bs;
  float maxoff;
  int PH, PW, tilesY, tilesX;
};

__device__ __forceinline__ void dcnv2_table2_body(const Tab2Args& a, int tile) {
  __shared__ int wbb[4][4];
  const int tid = threadIdx.x;
  const int tx = tile % a.tilesX, ty = (tile / a.tilesX) % a.tilesY, b = tile / (a.tilesX * a.tilesY);
  int bb[4] = {0x7fffffff, -0x7fffffff, 0x7fffffff, -0x7fffffff};      // this thread's y0 min, y0 + 1 max, x0 min, x0 + 1 max
  const int plane = a.Ho * a.Wo;
  int y0v[3], x0v[3]; bool live[3]; float4 wv[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = tid + 256 * i;
    live[i] = false; y0v[i] = 0; x0v[i] = 0; wv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e >= 576) continue;
    const int t = e >> 6, p = e & 63;
    const int py = p / a.PW, px = p - py * a.PW;
    const int ho = ty * a.PH + py, wo = tx * a.PW + px;
    if (py >= a.PH || ho >= a.Ho || wo >= a.Wo) continue;
    const int pix = ho * a.Wo + wo;
    const float* ob = a.off + (size_t)b * a.off_bs;
    float dy = ob[(size_t)(2 * t) * plane + pix], dx = ob[(size_t)(2 * t + 1) * plane + pix];
    float mod = 1.f;
    if (a.raw) {                                           // models/dcn.py:53-57: clamp(+-max_offset), 2 * sigmoid
      dy = fminf(fmaxf(dy, -a.maxoff), a.maxoff);
      dx = fminf(fmaxf(dx, -a.maxoff), a.maxoff);
      mod = 2.f / (1.f + expf(-ob[(size_t)(18 + t) * plane + pix]));
    } else if (a.msk) {
      mod = a.msk[(size_t)b * a.msk_bs + (size_t)t * plane + pix];
    }
    const int ki = t / 3, kj = t - ki * 3;
    const float y = (float)(ho * a.stride - a.pad + ki) + dy, x = (float)(wo * a.stride - a.pad + kj) + dx;
    const bool inside = (y > -1.f) && (y < (float)a.H) && (x > -1.f) && (x < (float)a.W);
    if (!inside) continue;
    const float fy = floorf(y), fx = floorf(x);
    const float ly = y - fy, lx = x - fx, hy = 1.f - ly, hx = 1.f - lx;
    y0v[i] = (int)fy; x0v[i] = (int)fx; live[i] = true;
    wv[i] = make_float4(mod * hy * hx, mod * hy * lx, mod * ly * hx, mod * ly * lx);
    bb[0] = min(bb[0], y0v[i]); bb[1] = max(bb[1], y0v[i] + 1);
    bb[2] = min(bb[2], x0v[i]); bb[3] = max(bb[3], x0v[i] + 1);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {                       // wave reduction, then the four waves through LDS (atomics on four LDS words: 31 us)
    bb[0] = min(bb[0], __shfl_xor(bb[0], o, 64)); bb[1] = max(bb[1], __shfl_xor(bb[1], o, 64));
    bb[2] = min(bb[2], __shfl_xor(bb[2], o, 64)); bb[3] = max(bb[3], __shfl_xor(bb[3], o, 64));
  }
  if ((tid & 63) == 0) { wbb[tid >> 6][0] = bb[0]; wbb[tid >> 6][1] = bb[1]; wbb[tid >> 6][2] = bb[2]; wbb[tid >> 6][3] = bb[3]; }
  __syncthreads();
  bb[0] = min(min(wbb[0][0], wbb[1][0]), min(wbb[2][0], wbb[3][0])); bb[1] = max(max(wbb[0][1], wbb[1][1]), max(wbb[2][1], wbb[3][1]));
  bb[2] = min(min(wbb[0][2], wbb[1][2]), min(wbb[2][2], wbb[3][2])); bb[3] = max(max(wbb[0][3], wbb[1][3]), max(wbb[2][3], wbb[3][3]));
  const bool any = bb[1] >= bb[0];
  const int wy0 = any ? bb[0] : 0, wx0 = any ? bb[2] : 0, wh = any ? bb[1] - bb[0] + 1 : 0, ww = any ? bb[3] - bb[2] + 1 : 0;
  const int fallback = (wh > WROWS || ww > WPITCH) ? 1 : 0;
  float4* tp = a.tab + (size_t)tile * TILE2_F4;
  if (tid == 0) {
    tp[0] = make_float4(__int_as_float(wy0), __int_as_float(wx0), __int_as_float(wh), __int_as_float(ww));
    tp[1] = make_float4(__int_as_float(fallback), 0.f, 0.f, 0.f);
  }
  const unsigned base = (unsigned)b * (unsigned)a.C * (unsigned)(a.H * a.W);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = tid + 256 * i;
    if (e >= 576) continue;
    unsigned meta = 0u, goff = 0u;
    if (live[i]) {
      const int y0 = y0v[i], x0 = x0v[i];
      const unsigned v00 = (y0 >= 0 && x0 >= 0) ? 1u : 0u, v01 = (y0 >= 0 && x0 + 1 <= a.W - 1) ? 2u : 0u;
      const unsigned v10 = (y0 + 1 <= a.H - 1 && x0 >= 0) ? 4u : 0u, v11 = (y0 + 1 <= a.H - 1 && x0 + 1 <= a.W - 1) ? 8u : 0u;
      const unsigned loff = fallback ? 0u : (unsigned)((y0 - wy0) * WPITCH + (x0 - wx0));
      meta = loff | ((v00 | v01 | v10 | v11) << 16);
      goff = (base + (unsigned)(y0 * a.W + x0)) * 4u;      // (wraps for y0 / x0 = -1: used only through the validity bits)
    }
    tp[2 + e] = make_float4(__uint_as_float(meta), __uint_as_float(goff), 0.f, 0.f);
    tp[2 + 576 + e] = wv[i];
  }
}

__global__ __launch_bounds__(256) void dcnv2_table_kernel(TabArgs a) { dcnv2_table_body(a, blockIdx.x); }
// both tables of a call in ONE launch: blocks [0, n1) the per-pixel gather table, the rest one patch each of the window table
__global__ __launch_bounds__(256) void dcnv2_tables_kernel(TabArgs a, Tab2Args t, int n1) {
  if ((int)blockIdx.x < n1) dcnv2_table_body(a, blockIdx.x);
  else dcnv2_table2_body(t, (int)blockIdx.x - n1);
}

struct Fwd2Args {
  const float* x; const float* w; const float* bias; const float4* tab; float* y; float* ws;
  int B, C, H, W, M, K, Ho, Wo;
  int PH, PW, tilesY, tilesX, tilesM, ptiles, nblocks, splits, epi;
  int xbytes, wbytes;
};

// Tile (64 * TM) output channels x one 64-pixel patch, 256 threads = 2 x 2 waves (wave tile 32 * TM x 32), K slices of 18 (two channels).
// Per iteration s: [global loads: weights of slice s + 1, windows of slice s + 2] -> [sample slice s + 1 from its windows (in LDS since the
// previous iteration) into Bs] -> [MFMA loop on slice s] -> [weights, windows -> LDS] -> ONE barrier.  The loop body is straight-line code:
// whatever a thread has no work for (weight pieces beyond the tile, window elements beyond the window, sampling rows 18 / 19 of waves 2 / 3)
// is a load at the out-of-range offset and a store to a dummy LDS word, never a branch (hipcc's waitcnt pass gives up across uniform branches).
// NWL: window elements per thread and channel (1: windows up to 256 elements, 2: 512, 5: WROWS x WPITCH); 0: the global-memory fallback.
#ifdef PRN_DCN2_TIMING
// Profiling aid (side build only, tools/dcn2_phase_timing.py): wave 0 of every workgroup sums the shader clocks it spends in the five phases of
// an iteration (load issue, sampling, MFMA loop, LDS stores, barrier) + iterations + entry / exit wall clock.
__device__ long long prn_dcn2_dbg[4096 * 10];
#define D2_STAMP(v_) const long long v_ = (long long)__builtin_readcyclecounter()
#else
#define D2_STAMP(v_) do { } while (0)
#endif

template <int TM> struct Fwd2Lds {
  static constexpr int BM = 64 * TM, LDA = W2_BK + 1;
  static constexpr int ABUF = BM * LDA + 2;             // + a dummy pair
  static constexpr int BBUF = (W2_BK + 2) * 64;         // + two dummy rows
  static constexpr int WBUF = 2 * WCH + 1;              // two channels + a dummy word
};

template <int TM, int NWL>
__device__ __forceinline__ void dcnv2_fwd2_body(const Fwd2Args& a, float* As, float* Bs, float* Win, int wy0, int wx0, int wh, int ww, int ptile, int mt) {
  using L = Fwd2Lds<TM>;
  constexpr int BM = L::BM, LDA = L::LDA;
  constexpr bool FALLBACK = NWL == 0;
  constexpr int NW = FALLBACK ? 1 : NWL;
  constexpr int NAL = (BM * 9 + 255) / 256;            // 8-byte weight pieces per thread per slice (a row's slice is 9 pieces)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = mt * BM;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.x, a.xbytes), wr = make_rsrc(a.w, a.wbytes);
  const int tpp = a.tilesY * a.tilesX, b = ptile / tpp;
  const int HW = a.H * a.W;

  // table entries of this thread's sampling rows r = wave + 4 * i (tap r % 9, channel r / 9 of the slice), pixel = lane; rows 18 / 19: weight 0
  const float4* tp = a.tab + (size_t)ptile * TILE2_F4;
  int soff[5]; unsigned goff[5], vbits[5]; float4 sw[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int r = wave + 4 * i, t = r % 9, cl = r / 9;
    const bool ok = r < W2_BK;
    const float4 mq = tp[2 + (ok ? t : 0) * 64 + lane];
    sw[i] = tp[2 + 576 + (ok ? t : 0) * 64 + lane];
    if (!ok) sw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned meta = __float_as_uint(mq.x);
    soff[i] = ok ? (int)(meta & 0xffffu) + cl * WCH : 0;
    vbits[i] = ok ? meta >> 16 : 0u;
    goff[i] = __float_as_uint(mq.y);
  }
  // weight pieces: e = tid + 256 * i -> (row e / 9, piece e % 9)
  unsigned aoff[NAL]; int alds[NAL];
#pragma unroll
  for (int i = 0; i < NAL; ++i) {
    const int e = tid + 256 * i, row = e / 9, pc = e - row * 9;
    aoff[i] = (row < BM && m0 + row < a.M) ? (unsigned)((m0 + row) * a.K + pc * 2) * 4u : OOB;
    alds[i] = row < BM ? row * LDA + pc * 2 : BM * LDA;
  }
  // window elements: e = tid + 256 * i -> (row e / ww, column e % ww); outside the image: zero (buffer range check)
  unsigned woff[NW]; int wlds[2][NW];
  if (!FALLBACK) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tid + 256 * i;
      const int r = ww > 0 ? e / ww : 0, j = e - r * ww;
      const int gy = wy0 + r, gx = wx0 + j;
      const bool in = e < wh * ww;
      woff[i] = (in && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? ((unsigned)b * (unsigned)a.C * (unsigned)HW + (unsigned)(gy * a.W + gx)) * 4u : OOB;
      wlds[0][i] = in ? r * WPITCH + j : 2 * WCH;
      wlds[1][i] = in ? WCH + r * WPITCH + j : 2 * WCH;
    }
  }

  float2 ra[NAL];
  float rw[2][NW];
  f32x16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int S = a.K / W2_BK;                               // slices (C is even: checked on the host)
  const int s0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * S / a.splits)), s1 = __builtin_amdgcn_readfirstlane((int)((blockIdx.y + 1) * S / a.splits));

  auto load_a = [&](int s) {
#pragma unroll
    for (int i = 0; i < NAL; ++i) ra[i] = bload2(wr, aoff[i], s * (W2_BK * 4));
  };
  auto store_a = [&](int buf) {
    float* A = As + buf * L::ABUF;
#pragma unroll
    for (int i = 0; i < NAL; ++i) { A[alds[i]] = ra[i].x; A[alds[i] + 1] = ra[i].y; }
  };
  auto load_w = [&](int s) {                               // windows of slice s (channels 2s, 2s + 1); past the end: nothing is fetched
    if (FALLBACK) return;
    const unsigned dead = s < s1 ? 0u : OOB;
#pragma unroll
    for (int cl = 0; cl < 2; ++cl)
#pragma unroll
      for (int i = 0; i < NW; ++i) rw[cl][i] = bload(xr, woff[i] | dead, (2 * s + cl) * HW * 4);
  };
  auto store_w = [&](int buf) {
    if (FALLBACK) return;
    float* Wb = Win + buf * L::WBUF;
#pragma unroll
    for (int cl = 0; cl < 2; ++cl)
#pragma unroll
      for (int i = 0; i < NW; ++i) Wb[wlds[cl][i]] = rw[cl][i];
  };
  auto sample = [&](int s, int buf) {                      // slice s -> Bs[buf] (rows wave + 4 i, pixel = lane)
    float* Bq = Bs + buf * L::BBUF;
    if (!FALLBACK) {
      const float* Wb = Win + (s & 1) * L::WBUF;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const float* q = Wb + soff[i];
        Bq[(wave + 4 * i) * 64 + lane] = (sw[i].x * q[0] + sw[i].y * q[1]) + (sw[i].z * q[WPITCH] + sw[i].w * q[WPITCH + 1]);
      }
    } else {
      float g[5][4];
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int r = wave + 4 * i;
        const int so = __builtin_amdgcn_readfirstlane((2 * s + (r >= 9 ? 1 : 0)) * HW * 4);
        g[i][0] = bload(xr, (vbits[i] & 1u) ? goff[i] : OOB, so);
        g[i][1] = bload(xr, (vbits[i] & 2u) ? goff[i] + 4u : OOB, so);
        g[i][2] = bload(xr, (vbits[i] & 4u) ? goff[i] + (unsigned)a.W * 4u : OOB, so);
        g[i][3] = bload(xr, (vbits[i] & 8u) ? goff[i] + (unsigned)a.W * 4u + 4u : OOB, so);
      }
#pragma unroll
      for (int i = 0; i < 5; ++i)
        Bq[(wave + 4 * i) * 64 + lane] = (sw[i].x * g[i][0] + sw[i].y * g[i][1]) + (sw[i].z * g[i][2] + sw[i].w * g[i][3]);
    }
  };
  auto mma = [&](int buf) {
    const float* A = As + buf * L::ABUF;
    const float* Bq = Bs + buf * L::BBUF;
#pragma unroll
    for (int kk = 0; kk < W2_BK / 2; ++kk) {
      float av[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = A[(wm * TM * 32 + i * 32 + (lane & 31)) * LDA + kk * 2 + (lane >> 5)];
      const float bv = Bq[(kk * 2 + (lane >> 5)) * 64 + wn * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv, acc[i], 0, 0, 0);
    }
  };

#ifdef PRN_DCN2_TIMING
  long long ph[5] = {0, 0, 0, 0, 0};
  const long long wall0 = (long long)wall_clock64();
#endif
  if (FALLBACK) {
    if (s0 < s1) {
      load_a(s0);
      store_a(s0 & 1);
      __syncthreads();
      sample(s0, s0 & 1);
      __syncthreads();
      for (int s = s0; s + 1 < s1; ++s) {
        load_a(s + 1);
        sample(s + 1, (s + 1) & 1);
        mma(s & 1);
        store_a((s + 1) & 1);
        __syncthreads();
      }
      mma((s1 - 1) & 1);
    }
  } else if (s0 < s1) {
    // Windowed path.  A wave's staging work is issued INSIDE its own MFMA loop, one piece per k step (a v_mfma_f32_32x32x2_f32 holds the
    // matrix pipe for 64 cycles, TM of them per k step: ~60 issue slots in their shadow) -- measured before: load issue 820 + sampling 1020 +
    // LDS stores 390 + barrier 330 cycles per iteration NEXT TO 2620 of MFMA loop, the co-resident workgroup hiding only part of it.
    // State at the top of iteration s: As[s & 1], Bs[s & 1] hold slice s; Win[(s + 1) & 1] the windows of slice s + 1; registers ra the
    // weights of slice s + 1, rw the windows of slice s + 2 (both loaded a whole iteration ago).  Group kk of the iteration:
    //   operands of k step kk + 1 -> registers; TM MFMAs of k step kk; weight piece kk: registers -> As[(s + 1) & 1], then its load for
    //   slice s + 2 into the same registers; even kk: one sampled element of slice s + 1 -> Bs[(s + 1) & 1]; kk = 1: windows -> Win[s & 1];
    //   kk = 3: window loads of slice s + 3.
    load_a(s0);
    load_w(s0);
    store_a(s0 & 1);
    store_w(s0 & 1);
    load_w(s0 + 1);
    store_w((s0 + 1) & 1);
    __syncthreads();
    sample(s0, s0 & 1);
    load_a(s0 + 1);
    load_w(s0 + 2);
    __syncthreads();
    for (int s = s0; s + 1 < s1; ++s) {
      D2_STAMP(c0);
      const float* A = As + (s & 1) * L::ABUF;
      const float* Bq = Bs + (s & 1) * L::BBUF;
      float* An = As + ((s + 1) & 1) * L::ABUF;
      float* Bn = Bs + ((s + 1) & 1) * L::BBUF;
      const float* Wb = Win + ((s + 1) & 1) * L::WBUF;
      float* Wn = Win + (s & 1) * L::WBUF;
      const unsigned wdead = s + 3 < s1 ? 0u : OOB;
      float av[2][TM], bv[2];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[0][i] = A[(wm * TM * 32 + i * 32 + (lane & 31)) * LDA + (lane >> 5)];
      bv[0] = Bq[(lane >> 5) * 64 + wn * 32 + (lane & 31)];
#pragma unroll
      for (int kk = 0; kk < W2_BK / 2; ++kk) {
        if (kk + 1 < W2_BK / 2) {
#pragma unroll
          for (int i = 0; i < TM; ++i) av[(kk + 1) & 1][i] = A[(wm * TM * 32 + i * 32 + (lane & 31)) * LDA + (kk + 1) * 2 + (lane >> 5)];
          bv[(kk + 1) & 1] = Bq[((kk + 1) * 2 + (lane >> 5)) * 64 + wn * 32 + (lane & 31)];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1][i], bv[kk & 1], acc[i], 0, 0, 0);
        // weight pieces i with i % 9 == kk (NAL <= 9: one piece per group; TM < 4 leaves some groups without)
#pragma unroll
        for (int i = 0; i < NAL; ++i)
          if (i == kk) {
            An[alds[i]] = ra[i].x; An[alds[i] + 1] = ra[i].y;
            ra[i] = bload2(wr, aoff[i], (s + 2) * (W2_BK * 4));
          }
        if ((kk & 1) == 0) {                                 // sampled element kk / 2 of slice s + 1
          const int e = kk >> 1;
          const float* q = Wb + soff[e];
          Bn[(wave + 4 * e) * 64 + lane] = (sw[e].x * q[0] + sw[e].y * q[1]) + (sw[e].z * q[WPITCH] + sw[e].w * q[WPITCH + 1]);
        }
        if (kk == 1) {
#pragma unroll
          for (int cl = 0; cl < 2; ++cl)
#pragma unroll
            for (int i = 0; i < NW; ++i) Wn[wlds[cl][i]] = rw[cl][i];
        }
        if (kk == 3) {
#pragma unroll
          for (int cl = 0; cl < 2; ++cl)
#pragma unroll
            for (int i = 0; i < NW; ++i) rw[cl][i] = bload(xr, woff[i] | wdead, (2 * (s + 3) + cl) * HW * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      D2_STAMP(c4);
      __syncthreads();
#ifdef PRN_DCN2_TIMING
      D2_STAMP(c5);
      ph[2] += c4 - c0; ph[4] += c5 - c4;
#endif
    }
    mma((s1 - 1) & 1);
  }
#ifdef PRN_DCN2_TIMING
  {
    const int wg = blockIdx.x + gridDim.x * blockIdx.y;
    if (tid == 0 && wg < 4096) {
      long long* o = prn_dcn2_dbg + wg * 10;
      for (int i = 0; i < 5; ++i) o[i] = ph[i];
      o[5] = s1 - s0 - 1; o[6] = wall0; o[7] = (long long)wall_clock64(); o[8] = NWL; o[9] = wh * 1000 + ww;
    }
  }
#endif

  // epilogue (C/D layout: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5))
  const int p = wn * 32 + (lane & 31);
  if (a.splits > 1) {                                      // partials [split][patch][M][64]
    float* outp = a.ws + ((size_t)blockIdx.y * a.ptiles + ptile) * (size_t)a.M * 64 + p;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mbase = m0 + wm * TM * 32 + i * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2);
        if (m < a.M) outp[(size_t)m * 64] = acc[i][r];
      }
    }
  } else {
    const int trem = ptile - b * tpp, ty = trem / a.tilesX, tx = trem - ty * a.tilesX;
    const int py = p / a.PW, px = p - py * a.PW;
    const int ho = ty * a.PH + py, wo = tx * a.PW + px;
    if (py < a.PH && ho < a.Ho && wo < a.Wo) {
      float* outp = a.y + (size_t)b * a.M * a.Ho * a.Wo + (size_t)ho * a.Wo + wo;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mbase = m0 + wm * TM * 32 + i * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + (r & 3) + 8 * (r >> 2);
          if (m >= a.M) continue;
          float v = acc[i][r];
          if (a.bias) v += a.bias[m];
          if (a.epi == PRN_EPI_RELU) v = fmaxf(v, 0.f);
          outp[(size_t)m * a.Ho * a.Wo] = v;
        }
      }
    }
  }
}

template <int TM>
__global__ __launch_bounds__(256, (TM == 4 ? 2 : 3)) void dcnv2_fwd2_kernel(Fwd2Args a) {
  using L = Fwd2Lds<TM>;
  __shared__ float As[2 * L::ABUF];
  __shared__ float Bs[2 * L::BBUF];
  __shared__ float Win[2 * L::WBUF];
  const int id = prn_xcd_remap(blockIdx.x, a.nblocks);
  const int mt = id % a.tilesM, ptile = id / a.tilesM;
  const float4* tp = a.tab + (size_t)ptile * TILE2_F4;
  const float4 h0 = tp[0], h1 = tp[1];
  const int wy0 = __builtin_amdgcn_readfirstlane(__float_as_int(h0.x)), wx0 = __builtin_amdgcn_readfirstlane(__float_as_int(h0.y));
  const int wh = __builtin_amdgcn_readfirstlane(__float_as_int(h0.z)), ww = __builtin_amdgcn_readfirstlane(__float_as_int(h0.w));
  const int fallback = __builtin_amdgcn_readfirstlane(__float_as_int(h1.x));
  // zero windows once: what a slice's loads do not cover (dead entries point at offset 0) must read as finite zeros in every slice
  for (int i = threadIdx.x; i < 2 * L::WBUF; i += 256) Win[i] = 0.f;
  __syncthreads();
  if (fallback) dcnv2_fwd2_body<TM, 0>(a, As, Bs, Win, wy0, wx0, wh, ww, ptile, mt);
  else if (wh * ww <= 256) dcnv2_fwd2_body<TM, 1>(a, As, Bs, Win, wy0, wx0, wh, ww, ptile, mt);
  else if (wh * ww <= 512) dcnv2_fwd2_body<TM, 2>(a, As, Bs, Win, wy0, wx0, wh, ww, ptile, mt);
  else dcnv2_fwd2_body<TM, 5>(a, As, Bs, Win, wy0, wx0, wh, ww, ptile, mt);
}

// y = epi(bias + sum over splits of the patch-major partials [split][patch][M][64])
__global__ __launch_bounds__(256) void dcnv2_patch_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias, float* __restrict__ y, int B, int M,
                                                                 int Ho, int Wo, int PH, int PW, int tilesY, int tilesX, int splits, int epi) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * M * Ho * Wo;
  if (i >= total) return;
  int wo, ho; int64_t bm;
  prn_idx3(i, Wo, Ho, wo, ho, bm);
  const int m = (int)(bm % M), b = (int)(bm / M);
  const int ty = ho / PH, py = ho - ty * PH, tx = wo / PW, px = wo - tx * PW;
  const int64_t ptiles = (int64_t)B * tilesY * tilesX, ptile = ((int64_t)b * tilesY + ty) * tilesX + tx;
  const float* q = ws + (ptile * M + m) * 64 + py * PW + px;
  float v = 0.f;
  for (int s = 0; s < splits; ++s) v += q[(size_t)s * ptiles * M * 64];
  if (bias) v += bias[m];
  if (epi == PRN_EPI_RELU) v = fmaxf(v, 0.f);
  y[i] = v;
}

// ------------------------------------------------------------------------------------------------ host side
int check_dcn(const prn_dcn_desc* d, const char* who) {
  PRN_REQUIRE(d != nullptr, "%s: null descriptor", who);
  PRN_REQUIRE(d->B > 0 && d->C > 0 && d->H > 0 && d->W > 0 && d->M > 0 && d->Ho > 0 && d->Wo > 0 && d->stride > 0 && d->pad >= 0, "%s: empty dimension", who);
  PRN_REQUIRE(d->Ho == (d->H + 2 * d->pad - 3) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - 3) / d->stride + 1,
              "%s: output size %dx%d does not match a 3x3 kernel with stride %d pad %d on %dx%d", who, d->Ho, d->Wo, d->stride, d->pad, d->H, d->W);
  PRN_REQUIRE((int64_t)d->B * d->C * d->H * d->W < (1LL << 29) && (int64_t)d->B * d->M * d->Ho * d->Wo < (1LL << 29) &&
              ((int64_t)d->M + 256) * d->C * 9 < (1LL << 29) && (int64_t)d->B * d->Ho * d->Wo * 9 * 32 < (1LL << 31),
              "%s: tensor larger than a buffer descriptor", who);
  return 0;
}
inline int npad(const prn_dcn_desc* d) { return cdiv((int64_t)d->B * d->Ho * d->Wo, 64) * 64; }

struct FPlan { int tm, splits; };
FPlan plan_dcn_fwd(int M, int N, int K) {
  static const prn_env4 forced_ = prn_env_ints("PRN_DCN_FWD");                        // PRN_DCN_FWD="tm,splits" (tuning sweeps)
  const int* forced = forced_.v;
  FPlan p;
  const int kt = cdiv(K, 16);
  if (forced[0] > 0) { p.tm = forced[0]; p.splits = forced[1] > 0 ? forced[1] : 1; }
  else {
    // Every output-channel tile gathers the SAME operand again, so the tallest tile that still fills the GPU wins:
    // candidates (tile height, K splits) are scored by how evenly tiles * splits spreads over 256 CUs x resident workgroups.
    double best = -1.0;
    p.tm = 1; p.splits = 1;
    const int tms[3] = {4, 2, 1}, res[3] = {3, 3, 4};
    for (int q = 0; q < 3; ++q) {
      const int tm = tms[q];
      if (tm > 1 && M <= 32 * tm) continue;
      const int64_t tiles = (int64_t)cdiv(M, 64 * tm) * cdiv(N, 64);
      for (int s = 1; s <= 8; ++s) {
        if (s > 1 && kt / s < 8) break;
        const double waves = (double)(tiles * s) / (256.0 * res[q]);
        double eff = waves / (double)((int64_t)(waves + 0.999999));
        if (waves < 1.0) eff = waves;
        const double score = eff * (1.0 - 0.03 * (s - 1)) * (tm == 4 ? 1.04 : (tm == 2 ? 1.02 : 1.0));
        if (score > best + 1e-9) { best = score; p.tm = tm; p.splits = s; }
      }
    }
  }
  if (p.splits > kt) p.splits = kt;
  if (p.splits < 1) p.splits = 1;
  return p;
}

// windowed forward: patch shape (fewest patches; the window of a patch must leave room for the offsets inside WROWS x WPITCH), tile height, K splits
struct F2Plan { int on, tm, tilesM, splits, PH, PW, tilesY, tilesX, ptiles; };
int dcn_v2_enabled() {
  static const int v = prn_env_int("PRN_DCN_V2", 1);                                     // PRN_DCN_V2=0: the gather kernels (A/B runs)
  return v;
}
F2Plan plan_dcn_fwd2(const prn_dcn_desc* d) {
  F2Plan p;
  p.on = dcn_v2_enabled() && (d->C % 2 == 0);            // (a K slice is two whole channels; an odd channel count keeps the gather kernel)
  const int cand[4][2] = {{8, 8}, {4, 16}, {16, 4}, {6, 10}};
  int best = -1;
  p.PH = 8; p.PW = 8;
  for (int q = 0; q < 4; ++q) {
    const int ph = cand[q][0], pw = cand[q][1];
    if ((ph - 1) * d->stride + 3 + 4 > WROWS || (pw - 1) * d->stride + 3 + 4 > WPITCH) continue;      // room for +-2 pixels of offset
    const int n = cdiv(d->Ho, ph) * cdiv(d->Wo, pw);
    if (best < 0 || n < best) { best = n; p.PH = ph; p.PW = pw; }
  }
  p.tilesY = cdiv(d->Ho, p.PH); p.tilesX = cdiv(d->Wo, p.PW); p.ptiles = d->B * p.tilesY * p.tilesX;
  p.tm = d->M > 128 ? 4 : (d->M > 64 ? 2 : 1);
  p.tilesM = cdiv(d->M, 64 * p.tm);
  static const prn_env4 forced_ = prn_env_ints("PRN_DCN_FWD2");                        // PRN_DCN_FWD2="tm,splits" (tuning sweeps)
  const int* forced = forced_.v;
  if (forced[0] > 0) { p.tm = forced[0]; p.tilesM = cdiv(d->M, 64 * p.tm); }
  const int S = d->C / 2, res = p.tm == 4 ? 2 : 3;
  const int64_t tiles = (int64_t)p.ptiles * p.tilesM;
  double bs = -1.0;
  p.splits = 1;
  for (int s = 1; s <= 8; ++s) {
    if (s > 1 && S / s < 8) break;
    const double waves = (double)(tiles * s) / (256.0 * res);
    double eff = waves / (double)((int64_t)(waves + 0.999999));
    if (waves < 1.0) eff = waves;
    const double score = eff * (1.0 - 0.08 * (s - 1));      // (a split costs a prologue, a partial tile and its share of the reduction: sweep in profiles/r05_*_dcn_fwd2_sweep.txt)
    if (score > bs + 1e-9) { bs = score; p.splits = s; }
  }
  if (forced[0] > 0 && forced[1] > 0) p.splits = forced[1] < S ? forced[1] : S;
  // patches that do not tile the map well leave lanes idle through the whole K loop (15 x 20 outputs: six 8 x 8 patches = 384 lanes for 300 pixels;
  // the gather kernel's linear 64-pixel tiles waste 1 %): such a layer keeps the gather kernel
  if ((int64_t)p.ptiles * 64 * 8 > (int64_t)d->B * d->Ho * d->Wo * 9 && forced[0] <= 0) p.on = 0;
  return p;
}
inline int64_t table1_bytes(const prn_dcn_desc* d) { return ((int64_t)npad(d) * 9 * 32 + 255) & ~255LL; }

struct WPlan { int tm, tilesM, tilesJ, splits, chunks; };
WPlan plan_dcn_wgrad(const prn_gemm_opts& o, int M, int K, int N) {
  static const prn_env4 forced_ = prn_env_ints("PRN_DCN_WGRAD");                        // PRN_DCN_WGRAD="tm,splits"
  const int* forced = forced_.v;
  WPlan p;
  p.tm = (forced[0] > 0) ? forced[0] : (M > 128 ? 4 : (M > 64 ? 2 : 1));   // every output-channel tile re-gathers the columns: tallest tile
  p.tilesM = cdiv(M, 64 * p.tm);
  p.tilesJ = cdiv(K, 64);
  p.chunks = cdiv(N, 16);
  const int tiles = p.tilesM * p.tilesJ;
  const int slots = 256 * (p.tm == 1 ? 4 : (p.tm == 2 ? 3 : 2));
  int s = tiles < slots ? 2 * slots / tiles : 1;
  if (o.wgrad_wgs > 0) s = tiles < o.wgrad_wgs ? o.wgrad_wgs / tiles : 1;     // deferred to the side stream: see plan_wgrad in prn_conv.hip
  int cap = p.chunks / 8 > 0 ? p.chunks / 8 : 1;          // at least 128 pixels per split
  const int sbw = (int)(0.0035 * (double)N) > 1 ? (int)(0.0035 * (double)N) : 1;   // workspace round trip stays small
  if (sbw < cap) cap = sbw;
  if (cap > 256) cap = 256;
  if (s > cap) s = prn_quantise_splits(tiles, cap);
  if (forced[0] > 0 && forced[1] > 0) s = forced[1] < p.chunks ? forced[1] : p.chunks;
  p.splits = s < 1 ? 1 : s;
  return p;
}

}  // namespace

extern "C" int64_t prn_dcnv2_table_bytes(const prn_dcn_desc* d) {
  if (check_dcn(d, "prn_dcnv2_table_bytes")) return -1;
  const F2Plan p2 = plan_dcn_fwd2(d);                      // [ per-pixel gather table (weight gradient) | per-patch window table (forward) ]
  return table1_bytes(d) + (p2.on ? (int64_t)p2.ptiles * TILE2_F4 * 16 : 0);
}

extern "C" int prn_dcnv2_table(const prn_dcn_desc* d, const float* offset, const float* mask, void* table, void* stream) {
  if (int e = check_dcn(d, "prn_dcnv2_table")) return e;
  PRN_REQUIRE(offset && table, "prn_dcnv2_table: null tensor");
  TabArgs a;
  a.off = offset; a.msk = d->raw ? nullptr : mask; a.tab = (float4*)table;
  a.B = d->B; a.C = d->C; a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo; a.stride = d->stride; a.pad = d->pad; a.raw = d->raw;
  const int64_t plane = (int64_t)d->Ho * d->Wo;
  a.off_bs = (d->raw ? 27 : 18) * plane; a.msk_bs = 9 * plane;
  a.maxoff = d->max_offset;
  a.N = d->B * d->Ho * d->Wo; a.Npad = npad(d);
  const F2Plan p2 = plan_dcn_fwd2(d);
  if (!p2.on) {
    hipLaunchKernelGGL(dcnv2_table_kernel, dim3(cdiv((int64_t)a.Npad * 9, 256)), dim3(256), 0, (hipStream_t)stream, a);
    PRN_CHECK_LAUNCH("prn_dcnv2_table");
  } else {
    Tab2Args t;
    t.off = offset; t.msk = a.msk; t.tab = (float4*)((char*)table + table1_bytes(d));
    t.B = d->B; t.C = d->C; t.H = d->H; t.W = d->W; t.Ho = d->Ho; t.Wo = d->Wo; t.stride = d->stride; t.pad = d->pad; t.raw = d->raw;
    t.off_bs = a.off_bs; t.msk_bs = a.msk_bs; t.maxoff = d->max_offset;
    t.PH = p2.PH; t.PW = p2.PW; t.tilesY = p2.tilesY; t.tilesX = p2.tilesX;
    const int n1 = cdiv((int64_t)a.Npad * 9, 256);
    hipLaunchKernelGGL(dcnv2_tables_kernel, dim3(n1 + p2.ptiles), dim3(256), 0, (hipStream_t)stream, a, t, n1);
    PRN_CHECK_LAUNCH("prn_dcnv2_table");
  }
  return 0;
}

extern "C" int64_t prn_dcnv2_fwd_ws_bytes(const prn_dcn_desc* d) {
  if (check_dcn(d, "prn_dcnv2_fwd_ws_bytes")) return -1;
  const F2Plan p2 = plan_dcn_fwd2(d);
  if (p2.on) return p2.splits > 1 ? (int64_t)p2.splits * p2.ptiles * 64 * d->M * 4 : 0;
  const FPlan p = plan_dcn_fwd(d->M, d->B * d->Ho * d->Wo, d->C * 9);
  return p.splits > 1 ? (int64_t)p.splits * d->B * d->M * d->Ho * d->Wo * 4 : 0;
}

extern "C" int prn_dcnv2_fwd_phase(const prn_dcn_desc* d, const float* x, const void* table, const float* w, const float* bias, float* y, void* ws,
                                   void* stream, int phase) {
  if (int e = check_dcn(d, "prn_dcnv2_fwd")) return e;
  PRN_REQUIRE(x && table && w && y, "prn_dcnv2_fwd: null tensor");
  FwdArgs a;
  a.x = x; a.w = w; a.bias = bias; a.tab = (const float4*)table; a.y = y; a.ws = (float*)ws;
  a.B = d->B; a.C = d->C; a.HW = d->H * d->W; a.M = d->M; a.K = d->C * 9; a.HoWo = d->Ho * d->Wo; a.N = d->B * a.HoWo;
  a.epi = d->epilogue;
  a.xbytes = d->B * d->C * a.HW * 4; a.wbytes = d->M * a.K * 4;
  const F2Plan p2 = plan_dcn_fwd2(d);
  if (p2.on) {
    Fwd2Args f;
    f.x = x; f.w = w; f.bias = bias; f.tab = (const float4*)((const char*)table + table1_bytes(d)); f.y = y; f.ws = (float*)ws;
    f.B = d->B; f.C = d->C; f.H = d->H; f.W = d->W; f.M = d->M; f.K = d->C * 9; f.Ho = d->Ho; f.Wo = d->Wo;
    f.PH = p2.PH; f.PW = p2.PW; f.tilesY = p2.tilesY; f.tilesX = p2.tilesX; f.tilesM = p2.tilesM; f.ptiles = p2.ptiles;
    f.nblocks = p2.ptiles * p2.tilesM; f.splits = p2.splits; f.epi = d->epilogue;
    f.xbytes = a.xbytes; f.wbytes = a.wbytes;
    PRN_REQUIRE(p2.splits == 1 || ws != nullptr, "prn_dcnv2_fwd: workspace required (%d K splits, see prn_dcnv2_fwd_ws_bytes)", p2.splits);
    PRN_REQUIRE((reinterpret_cast<uintptr_t>(w) & 7) == 0, "prn_dcnv2_fwd: w must be 8-byte aligned");
    hipStream_t st2 = (hipStream_t)stream;
    if (phase != 2) {
      const dim3 grid(f.nblocks, p2.splits), block(256);
      if (p2.tm == 4) hipLaunchKernelGGL((dcnv2_fwd2_kernel<4>), grid, block, 0, st2, f);
      else if (p2.tm == 2) hipLaunchKernelGGL((dcnv2_fwd2_kernel<2>), grid, block, 0, st2, f);
      else hipLaunchKernelGGL((dcnv2_fwd2_kernel<1>), grid, block, 0, st2, f);
      PRN_CHECK_LAUNCH("prn_dcnv2_fwd/windowed");
    }
    if (phase != 1 && p2.splits > 1) {
      const int64_t total = (int64_t)d->B * d->M * d->Ho * d->Wo;
      hipLaunchKernelGGL(dcnv2_patch_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st2, (const float*)ws, bias, y, d->B, d->M, d->Ho, d->Wo, p2.PH, p2.PW,
                         p2.tilesY, p2.tilesX, p2.splits, d->epilogue);
      PRN_CHECK_LAUNCH("prn_dcnv2_fwd/patch reduce");
    }
    return 0;
  }
  const FPlan p = plan_dcn_fwd(a.M, a.N, a.K);
  PRN_REQUIRE(p.splits == 1 || ws != nullptr, "prn_dcnv2_fwd: workspace required (%d K splits, see prn_dcnv2_fwd_ws_bytes)", p.splits);
  a.tilesM = cdiv(a.M, 64 * p.tm); a.nblocks = a.tilesM * cdiv(a.N, 64); a.splits = p.splits;
  hipStream_t st = (hipStream_t)stream;
  if (phase != 2) {
    const dim3 grid(a.nblocks, p.splits), block(256);
    const bool k4 = (a.K & 3) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0;
#define PRN_DCN_FWD(TM_) do { if (k4) hipLaunchKernelGGL((dcnv2_fwd_kernel<TM_, true>), grid, block, 0, st, a); \
                              else hipLaunchKernelGGL((dcnv2_fwd_kernel<TM_, false>), grid, block, 0, st, a); } while (0)
    if (p.tm == 4) {                                       // 256 x 64 tile: 8-deep K slices keep LDS at 3 workgroups per CU
      if (k4) hipLaunchKernelGGL((dcnv2_fwd_kernel<4, true, 8>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((dcnv2_fwd_kernel<4, false, 8>), grid, block, 0, st, a);
    } else if (p.tm == 2) PRN_DCN_FWD(2); else PRN_DCN_FWD(1);
#undef PRN_DCN_FWD
    PRN_CHECK_LAUNCH("prn_dcnv2_fwd");
  }
  if (phase != 1 && p.splits > 1)
    return prn_launch_reduce_epilogue((const float*)ws, bias, nullptr, y, (int64_t)a.B * a.M * a.HoWo, a.M, a.HoWo, p.splits, a.epi, st);
  return 0;
}

extern "C" int prn_dcnv2_fwd(const prn_dcn_desc* d, const float* x, const void* table, const float* w, const float* bias, float* y, void* ws,
                             void* stream) {
  return prn_dcnv2_fwd_phase(d, x, table, w, bias, y, ws, stream, 0);
}

extern "C" int64_t prn_dcnv2_bwd_weight_ws_bytes(const prn_dcn_desc* d) {
  if (check_dcn(d, "prn_dcnv2_bwd_weight_ws_bytes")) return -1;
  const WPlan p = plan_dcn_wgrad(d->opts, d->M, d->C * 9, d->B * d->Ho * d->Wo);
  return p.splits > 1 ? (int64_t)p.splits * d->M * d->C * 9 * 4 : 0;
}

extern "C" int prn_dcnv2_bwd_weight_phase(const prn_dcn_desc* d, const float* x, const void* table, const float* dy, float* dw, void* ws, void* stream,
                                          int phase) {
  if (int e = check_dcn(d, "prn_dcnv2_bwd_weight")) return e;
  PRN_REQUIRE(x && table && dy && dw, "prn_dcnv2_bwd_weight: null tensor");
  WgArgs a;
  a.x = x; a.dy = dy; a.tab = (const float4*)table;
  a.B = d->B; a.C = d->C; a.HW = d->H * d->W; a.M = d->M; a.K = d->C * 9; a.HoWo = d->Ho * d->Wo; a.N = d->B * a.HoWo;
  a.xbytes = d->B * d->C * a.HW * 4; a.dybytes = d->B * d->M * a.HoWo * 4; a.tabbytes = (int)((int64_t)npad(d) * 9 * 32);
  const WPlan p = plan_dcn_wgrad(d->opts, a.M, a.K, a.N);
  a.tilesM = p.tilesM; a.tilesJ = p.tilesJ; a.splits = p.splits; a.chunks = p.chunks;
  {
    static const int ilv = prn_env_int("PRN_DCN_WGRAD_ILV", 1);                                   // PRN_DCN_WGRAD_ILV=0: the phase-separated loop (A/B)
    a.ilv = ilv;
  }
  PRN_REQUIRE(p.splits == 1 || ws != nullptr, "prn_dcnv2_bwd_weight: workspace required (%d splits)", p.splits);
  a.out = p.splits > 1 ? (float*)ws : dw;
  hipStream_t st = (hipStream_t)stream;
  if (phase != 2) {
    const dim3 grid(p.tilesM * p.tilesJ, p.splits), block(256);
    const bool n4 = (a.HoWo & 3) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0;
#define PRN_DCN_WG(TM_) do { if (n4) hipLaunchKernelGGL((dcnv2_wgrad_kernel<TM_, true>), grid, block, 0, st, a); \
                             else hipLaunchKernelGGL((dcnv2_wgrad_kernel<TM_, false>), grid, block, 0, st, a); } while (0)
    if (p.tm == 4) PRN_DCN_WG(4); else if (p.tm == 2) PRN_DCN_WG(2); else PRN_DCN_WG(1);
#undef PRN_DCN_WG
    PRN_CHECK_LAUNCH("prn_dcnv2_bwd_weight");
  }
  if (phase != 1 && p.splits > 1) return prn_launch_reduce_splits((const float*)ws, dw, (int64_t)a.M * a.K, p.splits, st);
  return 0;
}

extern "C" int prn_dcnv2_bwd_weight(const prn_dcn_desc* d, const float* x, const void* table, const float* dy, float* dw, void* ws, void* stream) {
  return prn_dcnv2_bwd_weight_phase(d, x, table, dy, dw, ws, stream, 0);
}

#ifdef PRN_DCN2_TIMING
extern "C" int prn_debug_dcn2_timing(long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(prn_dcn2_dbg), sizeof(long long) * (size_t)(n < 4096 * 10 ? n : 4096 * 10)) == hipSuccess ? 0 : 1;
}
#endif

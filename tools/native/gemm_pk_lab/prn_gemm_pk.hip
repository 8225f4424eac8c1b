// Persistent strip GEMM on the gfx950 fp32 matrix cores for the plain-GEMM convolutions (1x1, stride 1) and the batched
// transform-domain products of the Winograd path:
//
//   Y_z[b][m][p] = epi( sum_k At_z[k][m] * X_z[b][k][p]  + bias[m] + addend_z[b][m][p] )        z < Z, b < B, p < HW
//
// Why a second GEMM kernel next to conv_igemm_kernel (prn_conv.hip).  The layers this serves are short: K = 64 .. 1024 on
// 2400 .. 9600 pixels, 30-70 us each, ~130 launches per training step.  With one workgroup per output tile such a launch is
// 1200 tiles on 1024 slots: a full round in lock step (every workgroup of the CU reaches its epilogue at the same moment),
// then a tail round on a fifth of the chip.  Here the grid is FIXED (2 workgroups per CU) and the work is cut to fit it:
//
//   * the unit of work is a 128-row x 32-pixel strip of the output (x a K range when K is split); all units of the launch
//     -- over the batch index z, the row tiles and the K splits -- are dealt out in contiguous, equal runs, one run per
//     workgroup, so the load per CU differs by at most one unit;
//   * a workgroup walks its run in chunks of up to four units (128 x 128) that share the weight panel; the four waves sit
//     side by side along the rows (32 x 32*TN each), so a narrow chunk shortens every wave alike;
//   * both operands are k-major ([k][row], [k][pixel]: the weights come transposed -- for a 1x1 layer that is the copy the
//     input gradient already uses, and vice versa), which makes each operand panel of a 16-deep K slice a set of 512-byte
//     rows: they go HBM/L2 -> LDS by `buffer_load_dwordx4 ... lds` (no staging registers, no LDS write pass), NS slices in
//     flight, one barrier per slice, counted vmcnt;
//   * the two workgroups of a CU drift apart, so one's epilogue (LDS transpose, float4 stores) runs under the other's MFMAs;
//   * XCD placement: workgroup g runs on XCD g % 8 and takes its run from that XCD's eighth of the pixel columns, for all row
//     tiles -- the activations of those columns are fetched into that L2 once.
//
// K split (S > 1): partial sums go to the workspace slice of the split, the fixed-order sum + epilogue is
// prn_launch_reduce_epilogue (same as conv_igemm_kernel's).  Without a split each output element is one k-ordered fmaf chain
// from k = 0 -- the same value conv_igemm_kernel produces unsplit.
#include <stdlib.h>
#include <type_traits>
#include "prn_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr unsigned OOB = 0x80000000u;   // byte offset no buffer descriptor covers: the load returns 0 without a memory access

struct PkArgs {
  const float* at; const float* x; const float* bias; const float* addend; float* y; float* ws;
  int M, K, B, HW, N, lda, Z, epi;
  int S, KT;              // K splits, 16-deep K slices
  int UN, tilesM;         // 32-pixel units per z, 128-row tiles
  int xbytes, abytes;     // descriptor ranges of one z of x / at
  long long zat, zx, zy;  // element strides per z
  long long slice;        // elements of one partial slice (S > 1): Z * B * M * HW
  int dbg;                     // lab only: 1 = no epilogue stores
  unsigned long long* trace;   // lab only (TRACE instances): per workgroup 64 timestamps of wave 0
};

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// raw buffer descriptor (stride 0, range `bytes`): words as __builtin_amdgcn_make_buffer_rsrc lays them out, kept as a plain
// vector so that it can be an "s" operand of the LDS-DMA statement below
__device__ __forceinline__ i32x4 make_desc(const void* p, int bytes) {
  const unsigned long long q = (unsigned long long)p;
  i32x4 d;
  d.x = __builtin_amdgcn_readfirstlane((int)(unsigned)q);
  d.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(q >> 32) & 0xffff);
  d.z = __builtin_amdgcn_readfirstlane(bytes);
  d.w = 0x00020000;
  return d;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// LDS-DMA: 64 lanes x 16 bytes from (descriptor, per-lane byte offset + scalar offset) to LDS [lds_byte_addr + 16*lane], no
// staging registers.  As an asm statement on purpose: hipcc then neither counts it in its own s_waitcnt bookkeeping (it would
// drain vmcnt(0) in front of every later LDS read of the same array -- e.g. the epilogue's transpose reads, which would
// thereby also wait for the epilogue's own stores) nor needs to know M0.  Completion is counted by hand (wait_vm).
__device__ __forceinline__ void lds_dma16(unsigned lds_byte_addr, i32x4 desc, unsigned voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane((int)lds_byte_addr)), "v"(voff), "s"(desc),
               "s"(__builtin_amdgcn_readfirstlane(soff))
               : "memory");
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One workgroup, one chunk: rows [m0, m0+128) x pixels [n0, n0 + 32*TN) of batch entry z, K slices [kt0, kt1).
template <int NS, int TN, bool TRACE>
__device__ __forceinline__ void run_chunk(const PkArgs& a, float* smem, int s, int m0, int z, int n0, int kt0, int kt1, unsigned long long* tr) {
  unsigned long long t_wait = 0;
  if (TRACE) tr[0] = __builtin_amdgcn_s_memtime();
  constexpr int PANEL = 16 * 128;            // floats per operand panel of one stage
  constexpr int STAGE = 2 * PANEL;
  const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const i32x4 ar = make_desc(a.at + (size_t)z * a.zat, a.abytes), xr = make_desc(a.x + (size_t)z * a.zx, a.xbytes);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;   // LDS byte address of the ring

  // this lane's part of the two panels: rows 4w + 2i + hi (i = 0, 1), four consecutive columns starting at 4*lo
  const int c4 = lo * 4;
  unsigned va[2], vb[2];
  int ka[2];
  {
    const int n = n0 + c4;
    const bool bok = c4 < 32 * TN && n < a.N;
    const int b = bok ? n / a.HW : 0, p = n - b * a.HW;
    const bool aok = m0 + c4 < a.M;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 4 * w + 2 * i + hi;
      ka[i] = a.K - r;                                            // row r of the slice at k0 exists iff k0 < K - r
      va[i] = aok ? (unsigned)(r * a.lda + m0 + c4) * 4u : OOB;
      vb[i] = bok ? (unsigned)((b * a.K + r) * a.HW + p) * 4u : OOB;
    }
  }
  auto issue = [&](int kt, int stage) {
    const int k0 = kt * 16;
    const unsigned st = lds0 + (unsigned)(stage * STAGE + 4 * w * 128) * 4u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool kok = k0 < ka[i];
      lds_dma16(st + i * 1024, ar, kok ? va[i] : OOB, k0 * a.lda * 4);
      lds_dma16(st + PANEL * 4 + i * 1024, xr, kok ? vb[i] : OOB, k0 * a.HW * 4);
    }
  };

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int nk = kt1 - kt0;
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (i < nk) issue(kt0 + i, i);

  const float* abase = smem + hi * 128 + 32 * w + lo;
  const float* bbase = smem + PANEL + hi * 128 + lo;
  int stage = 0;
  for (int t = 0; t < nk; ++t) {
    // slice t has landed (this wave's pieces, then everybody's); the stage of slice t-1 is free again after the barrier
    const int rem = nk - 1 - t;
    unsigned long long t0 = 0;
    if (TRACE) t0 = __builtin_amdgcn_s_memtime();
    if (NS >= 4 && rem >= 2) wait_vm<8>();
    else if (NS >= 3 && rem >= 1) wait_vm<4>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (TRACE) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); t_wait += t1 - t0; if (t == 0) tr[1] = t1; }
    if (t + NS - 1 < nk) issue(kt0 + t + NS - 1, stage == 0 ? NS - 1 : stage - 1);
    const float* as = abase + stage * STAGE;
    const float* bs = bbase + stage * STAGE;
    // all fragments of the slice first (8 + 8*TN registers), then 8*TN MFMAs back to back: the LDS latency is paid once per
    // slice, under the other workgroup's MFMAs
    float av[8], bv[8][TN];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      av[kk] = as[kk * 256];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[kk][j] = bs[kk * 256 + 32 * j];
    }
    __builtin_amdgcn_sched_barrier(0);          // (hipcc otherwise sinks each read to just above its MFMA: lgkmcnt(0) every two MFMAs)
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk][j], acc[j], 0, 0, 0);
    stage = stage + 1 == NS ? 0 : stage + 1;
  }

  if (TRACE) { tr[2] = __builtin_amdgcn_s_memtime(); tr[3] = t_wait; }
  // epilogue.  An accumulator block holds 16 rows of ONE pixel per lane; through a 16 x 36 LDS pad per wave (two passes of
  // eight registers) a lane gets four consecutive pixels of one row: float4 stores, 128 contiguous bytes per 8 lanes.
  // Branch-free: rows / pixels that do not exist carry the OOB offset (buffer stores drop them, buffer loads return 0).
  float* pad = smem + NS * STAGE + w * (16 * 36);
  const bool partial = a.S > 1;
  const int ybytes = a.B * a.M * a.HW * 4;
  const __amdgpu_buffer_rsrc_t yr = make_rsrc((partial ? a.ws + (size_t)s * a.slice : a.y) + (size_t)z * a.zy, ybytes);
  const bool has_add = !partial && a.addend != nullptr;
  const __amdgpu_buffer_rsrc_t addr = make_rsrc(has_add ? a.addend + (size_t)z * a.zy : a.y, has_add ? ybytes : 0);
  const __amdgpu_buffer_rsrc_t br = make_rsrc(a.bias ? a.bias : a.y, (!partial && a.bias) ? a.M * 4 : 0);
  const int epi = partial ? PRN_EPI_NONE : a.epi;
  const int crow = lane >> 3, ccol = (lane & 7) * 4;
  const int mrow = m0 + 32 * w + crow;                    // this lane's rows: mrow + 16h + 8q
  float bm[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int q = 0; q < 2; ++q) bm[h][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, (mrow + 16 * h + 8 * q) * 4, 0, 0));
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + 32 * j + ccol;
    const bool nok = n < a.N;
    const int b = nok ? n / a.HW : 0, p = n - b * a.HW;
    const unsigned base = (unsigned)((b * a.M + mrow) * a.HW + p) * 4u;
    f32x4 v[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int r = 0; r < 8; ++r) pad[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36 + lo] = acc[j][8 * h + r];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 2; ++q) v[h][q] = *reinterpret_cast<const f32x4*>(&pad[(crow + 8 * q) * 36 + ccol]);
      __builtin_amdgcn_wave_barrier();
    }
    unsigned off[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 2; ++q) off[h][q] = (nok && mrow + 16 * h + 8 * q < a.M) ? base + (unsigned)((16 * h + 8 * q) * a.HW) * 4u : OOB;
    if (has_add) {
      f32x4 t[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) t[h][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(addr, (int)off[h][q], 0, 0));
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) v[h][q] += t[h][q];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x4 o = v[h][q] + bm[h][q];
        if (epi == PRN_EPI_RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        else if (epi == PRN_EPI_SIGMOID) { o.x = 1.f / (1.f + __expf(-o.x)); o.y = 1.f / (1.f + __expf(-o.y)); o.z = 1.f / (1.f + __expf(-o.z)); o.w = 1.f / (1.f + __expf(-o.w)); }
        if (a.dbg != 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yr, (int)off[h][q], 0, 0);
      }
  }
  __builtin_amdgcn_s_barrier();           // every wave is done with the ring before the next chunk's first loads land in it
  if (TRACE) { tr[4] = __builtin_amdgcn_s_memtime(); tr[5] = ((unsigned long long)TN << 32) | (unsigned)nk; }
}

template <int NS, bool TRACE = false>
__global__ __launch_bounds__(256, 2) void gemm_pk_kernel(PkArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[NS * 2 * 16 * 128 + 4 * 16 * 36];
  // workgroup g -> XCD x = g % 8 (observed placement; speed only), j-th of J workgroups there
  const int g = blockIdx.x, G = gridDim.x;
  const int x = g & 7, j = g >> 3, J = G >> 3;
  const int CT = a.Z * a.UN;                                     // unit columns of the launch: (z, 32-pixel unit)
  const int cx0 = (int)((long long)x * CT / 8), cx1 = (int)((long long)(x + 1) * CT / 8), nx = cx1 - cx0;
  unsigned long long* tr = nullptr;
  int nchunk = 0;
  if (TRACE) {
    tr = a.trace + (size_t)g * 64;
    if (threadIdx.x == 0) {
      tr[0] = __builtin_amdgcn_s_memtime();
      tr[1] = wall_clock64();
      unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      tr[2] = ((unsigned long long)xcc << 32) | hw;
    }
  }
  if (nx <= 0) return;
  const long long T = (long long)a.S * a.tilesM * nx;            // units of this XCD: (split, row tile) major, column minor
  long long q = (long long)j * T / J;
  const long long q1 = (long long)(j + 1) * T / J;
  while (q < q1) {
    const int sm = (int)(q / nx), c = (int)(q - (long long)sm * nx);
    const int s = sm / a.tilesM, mt = sm - s * a.tilesM;
    const int col = cx0 + c, z = col / a.UN, u = col - z * a.UN;
    long long run = q1 - q;
    if (run > nx - c) run = nx - c;
    if (run > a.UN - u) run = a.UN - u;
    const int nch = ((int)run + 3) >> 2;
    int tn = ((int)run + nch - 1) / nch;                           // balanced chunk widths: 5 -> 3 + 2, 9 -> 3 + 3 + 3
    const int kt0 = (int)((long long)s * a.KT / a.S), kt1 = (int)((long long)(s + 1) * a.KT / a.S);
    const int m0 = mt * 128, n0 = u * 32;
    unsigned long long dummy[6];
    unsigned long long* ct = (TRACE && threadIdx.x == 0 && nchunk < 9) ? tr + 8 + 6 * nchunk : dummy;
    if (tn == 4) run_chunk<NS, 4, TRACE>(a, smem, s, m0, z, n0, kt0, kt1, ct);
    else if (tn == 3) run_chunk<NS, 3, TRACE>(a, smem, s, m0, z, n0, kt0, kt1, ct);
    else if (tn == 2) run_chunk<NS, 2, TRACE>(a, smem, s, m0, z, n0, kt0, kt1, ct);
    else run_chunk<NS, 1, TRACE>(a, smem, s, m0, z, n0, kt0, kt1, ct);
    q += tn;
    ++nchunk;
  }
  if (TRACE && threadIdx.x == 0) { tr[3] = __builtin_amdgcn_s_memtime(); tr[4] = wall_clock64(); tr[5] = nchunk; }
}

// ------------------------------------------------------------------------------------------------------------------------
// Version 2: ONE continuous slice stream per workgroup.  The trace of version 1 (tools/native/gemm_lab.cpp, LAB_TRACE=1) on
// 1x1 256->1024 @30x40: K loops at 88-95 % of the MFMA issue floor, but 3.5 k cycles per chunk waiting for its first slice and
// 6-14 k in its epilogue, and the two workgroups of a CU are in those phases at the same time (they start together and have the
// same chunk structure) -- a third of the launch.  Removing the stores altogether gained 2 us of 58: it is not the write
// burst, it is serialisation inside each wave.  So here nothing of a chunk's bookkeeping is a phase of its own:
//   * the LDS-DMA stream runs four slices ahead of the MFMAs ACROSS chunk boundaries (the next chunk's first slices are in
//     flight while this one finishes);
//   * the barrier of slice t+1 sits in the MIDDLE of slice t's MFMA stream; the fragments of an 8-step slice live in one
//     register set that is refilled half by half (steps 4-7 of slice t while steps 0-3 run, steps 0-3 of slice t+1 while
//     4-7 run), so no LDS latency is exposed inside a chunk; the DMA issues sit between MFMA groups;
//   * the epilogue is write-behind: the finished accumulators move to a second register set and are transposed / stored one
//     8-register piece per slice between the MFMAs of the NEXT chunk; only a workgroup's last chunk is flushed in the open.
// Chunks are 1-3 units wide (with a fourth unit the two accumulator sets, 2 x 64 registers, spill).
struct Chunk { int s, m0, z, n0, tn, kt0, nk; };   // nk == 0: none
struct Walker { long long q, q1; int nx, cx0; };

__device__ __forceinline__ Chunk next_chunk(const PkArgs& a, Walker& wk) {
  Chunk c;
  c.s = c.m0 = c.z = c.n0 = c.tn = c.kt0 = c.nk = 0;
  if (wk.q >= wk.q1) return c;
  const int sm = (int)(wk.q / wk.nx), col0 = (int)(wk.q - (long long)sm * wk.nx);
  c.s = sm / a.tilesM;
  const int mt = sm - c.s * a.tilesM;
  const int col = wk.cx0 + col0;
  c.z = col / a.UN;
  const int u = col - c.z * a.UN;
  long long run = wk.q1 - wk.q;
  if (run > wk.nx - col0) run = wk.nx - col0;
  if (run > a.UN - u) run = a.UN - u;
  { const int nch = ((int)run + 2) / 3; c.tn = ((int)run + nch - 1) / nch; }      // balanced widths <= 3: 4 -> 2 + 2, 5 -> 3 + 2, 7 -> 3 + 2 + 2
  c.kt0 = (int)((long long)c.s * a.KT / a.S);
  c.nk = (int)((long long)(c.s + 1) * a.KT / a.S) - c.kt0;
  c.m0 = mt * 128;
  c.n0 = u * 32;
  wk.q += c.tn;
  return c;
}

// s_waitcnt vmcnt for `ahead` (0..3) LDS-DMA groups of four that may stay in flight plus `young` epilogue stores issued after
// the group being waited for; rounded DOWN to a few encodable cases (waiting for more is always safe)
__device__ __forceinline__ void wait_vm_dyn(int ahead, int young) {
  if (ahead >= 2) { if (young >= 6) wait_vm<14>(); else wait_vm<8>(); }
  else if (ahead == 1) wait_vm<4>();
  else wait_vm<0>();
}

template <bool TRACE>
__global__ __launch_bounds__(256, 2) void gemm_pk2_kernel(PkArgs a) {
  constexpr int NS = 4, PANEL = 16 * 128, STAGE = 2 * PANEL;
  __shared__ __attribute__((aligned(16))) float smem[NS * STAGE + 4 * 16 * 36];
  const int g = blockIdx.x, G = gridDim.x;
  Walker wk;
  {
    const int x = g & 7, j = g >> 3, J = G >> 3;
    const int CT = a.Z * a.UN;
    wk.cx0 = (int)((long long)x * CT / 8);
    wk.nx = (int)((long long)(x + 1) * CT / 8) - wk.cx0;
    const long long T = (long long)a.S * a.tilesM * wk.nx;
    wk.q = wk.nx > 0 ? (long long)j * T / J : 0;
    wk.q1 = wk.nx > 0 ? (long long)(j + 1) * T / J : 0;
  }
  unsigned long long* tr = nullptr;
  unsigned long long trc[5] = {0, 0, 0, 0, 0};
  if (TRACE) {
    tr = a.trace + (size_t)g * 64;
    if (threadIdx.x == 0) { tr[0] = __builtin_amdgcn_s_memtime(); tr[1] = wall_clock64(); }
  }
  Chunk cur = next_chunk(a, wk);
  if (cur.nk == 0) return;
  Chunk nxt = next_chunk(a, wk);

  const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  const bool partial = a.S > 1;
  const long long nY = (long long)(a.Z - 1) * a.zy + (long long)a.B * a.M * a.HW;            // elements of one output (partial: one slice)
  const i32x4 ar = make_desc(a.at, (int)(((long long)(a.Z - 1) * a.zat + (long long)a.K * a.lda) * 4));
  const i32x4 xr = make_desc(a.x, (int)(((long long)(a.Z - 1) * a.zx + (long long)a.K * a.N) * 4));
  const __amdgpu_buffer_rsrc_t yr = make_rsrc(partial ? a.ws : a.y, (int)((partial ? a.slice * a.S : nY) * 4));
  const bool has_add = !partial && a.addend != nullptr;
  const __amdgpu_buffer_rsrc_t addr = make_rsrc(has_add ? a.addend : a.y, has_add ? (int)(nY * 4) : 0);
  const __amdgpu_buffer_rsrc_t br = make_rsrc(a.bias ? a.bias : a.y, (!partial && a.bias) ? a.M * 4 : 0);
  const float lowest = (!partial && a.epi == PRN_EPI_RELU) ? 0.f : -__builtin_inff();     // ReLU or nothing (sigmoid epilogues stay on version 1)

  // ---- load side.  BRANCH-FREE by construction: four LDS-DMA issues every slice, for ever; when the stream crosses into the next
  // chunk the lane offsets / scalar offsets are swapped in by selects (the next chunk's are prepared when that chunk becomes known),
  // and behind the last chunk they are OOB, i.e. the issues become no-ops that move nothing -- so the number of loads in flight is a
  // constant and s_waitcnt vmcnt takes an immediate.  (tools/native/issue_cost.cpp: an LDS-DMA, a buffer load, a ds_write or an
  // s_barrier between two MFMAs costs the wave nothing; what cost versions 2 / 3 of this kernel 10-25 % were TAKEN BRANCHES in the
  // slice loop -- a wave refetches its instruction stream after each, and the loop had ~10 of them per slice.)
  // This lane's pieces of the two panels are rows 4w + 2i + hi (i = 0, 1), columns 4*lo .. 4*lo + 3.
  const int c4 = lo * 4;
  int ka[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) ka[i] = a.K - (4 * w + 2 * i + hi);
  unsigned va[2], vb[2], va_n[2], vb_n[2];
  int Lk, Lkend, LzA, LzB, Lk_n, Lkend_n, LzA_n, LzB_n, ls = 0;
  auto lane_offsets = [&](const Chunk& c, unsigned (&oa)[2], unsigned (&ob)[2], int& k, int& kend, int& zA, int& zB) __attribute__((always_inline)) {
    const bool live = c.nk != 0;
    const int n = c.n0 + c4;
    const bool bok = live && c4 < 32 * c.tn && n < a.N;
    const int b = bok ? n / a.HW : 0, p = n - b * a.HW;
    const bool aok = live && c.m0 + c4 < a.M;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 4 * w + 2 * i + hi;
      oa[i] = aok ? (unsigned)(r * a.lda + c.m0 + c4) * 4u : OOB;
      ob[i] = bok ? (unsigned)((b * a.K + r) * a.HW + p) * 4u : OOB;
    }
    k = c.kt0; kend = live ? c.kt0 + c.nk : 0x7fffffff;           // (behind the last chunk: never crosses again)
    zA = (int)(c.z * a.zat * 4); zB = (int)(c.z * a.zx * 4);
  };
  auto issue_slice = [&]() __attribute__((always_inline)) {
    const int k0 = Lk * 16;
    const unsigned st = lds0 + (unsigned)(ls * STAGE + 4 * w * 128) * 4u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool kok = k0 < ka[i];
      lds_dma16(st + i * 1024, ar, kok ? va[i] : OOB, LzA + k0 * a.lda * 4);
      lds_dma16(st + PANEL * 4 + i * 1024, xr, kok ? vb[i] : OOB, LzB + k0 * a.HW * 4);
    }
    ls = (ls + 1) & (NS - 1);
    ++Lk;
    const bool cross = Lk == Lkend;                    // selects, not a branch
#pragma unroll
    for (int i = 0; i < 2; ++i) { va[i] = cross ? va_n[i] : va[i]; vb[i] = cross ? vb_n[i] : vb[i]; }
    Lk = cross ? Lk_n : Lk; LzA = cross ? LzA_n : LzA; LzB = cross ? LzB_n : LzB;
    Lkend = cross ? Lkend_n : Lkend;
  };

  // ---- compute side
  f32x16 acc[3], accP[3];
  const float* abase = smem + hi * 128 + 32 * w + lo;
  const float* bbase = smem + PANEL + hi * 128 + lo;
  int cs = 0;                                         // stage of the slice being computed
  float* pad = smem + NS * STAGE + w * (16 * 36);
  const int crow = lane >> 3, ccol = (lane & 7) * 4;
  unsigned pbase[3] = {OOB, OOB, OOB};           // write-behind state of the previous chunk; OOB: nothing to store
  int mrowP = 0;
  float bm[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accP[j][r] = 0.f;

  lane_offsets(cur, va, vb, Lk, Lkend, LzA, LzB);
  lane_offsets(nxt, va_n, vb_n, Lk_n, Lkend_n, LzA_n, LzB_n);
  issue_slice(); issue_slice(); issue_slice();
  if (a.dbg > 1 && g >= (G >> 1))                      // lab: start the second workgroup of every CU late by dbg x 64 cycles
    for (int i = 0; i < a.dbg; ++i) __builtin_amdgcn_s_sleep(1);

#define SB __builtin_amdgcn_sched_barrier(0)
  // Write-behind epilogue of the previous chunk, piece (j, h) = rows 16h .. 16h + 15 of column block j: eight accumulator registers
  // through the wave's 16 x 36 LDS pad come back as two float4 rows -> bias / addend / ReLU -> 16-byte stores, 128 contiguous bytes
  // per 8 lanes.  Always all six pieces: column blocks the previous chunk did not have carry the OOB offset and store nothing.
  auto piece = [&](auto jt, auto ht) __attribute__((always_inline)) {
    constexpr int j = decltype(jt)::value, h = decltype(ht)::value;
#pragma unroll
    for (int r = 0; r < 8; ++r) pad[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36 + lo] = accP[j][8 * h + r];
    __builtin_amdgcn_wave_barrier();
    f32x4 v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) v[q] = *reinterpret_cast<const f32x4*>(&pad[(crow + 8 * q) * 36 + ccol]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned off = (mrowP + 16 * h + 8 * q < a.M) ? pbase[j] + (unsigned)((16 * h + 8 * q) * a.HW) * 4u : OOB;
      f32x4 o = v[q];
      if (has_add) o += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(addr, (int)off, 0, 0));
      o += bm[h][q];
      o.x = fmaxf(o.x, lowest); o.y = fmaxf(o.y, lowest); o.z = fmaxf(o.z, lowest); o.w = fmaxf(o.w, lowest);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yr, (int)off, 0, 0);
    }
  };
  // the finished chunk becomes the "previous" one: accumulators to the second set, output offsets and bias of its rows
  auto retire = [&](auto tn, const Chunk& c) __attribute__((always_inline)) {
    constexpr int TN = decltype(tn)::value;
    mrowP = c.m0 + 32 * w + crow;
    const long long zoff = (long long)c.z * a.zy + (partial ? (long long)c.s * a.slice : 0);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j < TN) {
        accP[j] = acc[j];
        const int n = c.n0 + 32 * j + ccol;
        const bool nok = n < a.N;
        const int b = nok ? n / a.HW : 0, p = n - b * a.HW;
        pbase[j] = nok ? (unsigned)(zoff + (long long)(b * a.M + mrowP) * a.HW + p) * 4u : OOB;
      } else {
        pbase[j] = OOB;
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 2; ++q) bm[h][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, (mrowP + 16 * h + 8 * q) * 4, 0, 0));
  };

  // One slice: [slice landed: vmcnt, barrier] [four LDS-DMA issues for the slice three ahead] [optional write-behind piece]
  // [all fragment reads] [8 * TN MFMAs back to back].  YOUNG = epilogue stores issued since the group being waited for was (an
  // immediate: two loads groups of four always stay in flight behind it).
  auto slice = [&](auto tn, auto young, auto pj, auto ph) __attribute__((always_inline)) {
    constexpr int TN = decltype(tn)::value, J = decltype(pj)::value;
    unsigned long long tq0 = 0, tq1 = 0;
    if (TRACE) tq0 = __builtin_amdgcn_s_memtime();
    wait_vm<8 + decltype(young)::value>();
    if (TRACE) tq1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    if (TRACE) { const unsigned long long tq2 = __builtin_amdgcn_s_memtime(); trc[0] += tq1 - tq0; trc[1] += tq2 - tq1; trc[3] += 1; trc[4] += TN; if (J >= 0) trc[2] += 1; }
    issue_slice();
    if constexpr (J >= 0) piece(pj, ph);
    const float* as = abase + cs * STAGE;
    const float* bs = bbase + cs * STAGE;
    float av[8], bv[8][TN];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      av[kk] = as[kk * 256];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[kk][j] = bs[kk * 256 + 32 * j];
    }
    SB;                                                // (hipcc otherwise sinks each read to just above its MFMA: lgkmcnt(0) every two MFMAs)
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk][j], acc[j], 0, 0, 0);
    SB;
    cs = (cs + 1) & (NS - 1);
  };
  using N1 = std::integral_constant<int, -1>;
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  using Y0 = std::integral_constant<int, 0>; using Y2 = std::integral_constant<int, 2>; using Y4 = std::integral_constant<int, 4>; using Y6 = std::integral_constant<int, 6>;
  // all slices of the current chunk: the first six carry the six pieces of the previous chunk (straight-line code, the piece of
  // each slice is a compile-time one), the rest is the bare loop.  nk >= 4 (host); chunks shorter than six slices flush the
  // remaining pieces behind their last slice.
  auto kloop = [&](auto tn) __attribute__((always_inline)) {
    constexpr int TN = decltype(tn)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int nk = cur.nk;
    slice(tn, Y0{}, I0{}, I0{});
    slice(tn, Y2{}, I0{}, I1{});
    slice(tn, Y4{}, I1{}, I0{});
    slice(tn, Y6{}, I1{}, I1{});
    if (__builtin_expect(nk >= 6, 1)) {
      slice(tn, Y6{}, I2{}, I0{});
      slice(tn, Y6{}, I2{}, I1{});
#pragma nounroll
      for (int t = 6; t < nk; ++t) slice(tn, Y0{}, N1{}, N1{});
    } else {
#pragma nounroll
      for (int t = 4; t < nk; ++t) slice(tn, Y0{}, N1{}, N1{});
      piece(I2{}, I0{}); piece(I2{}, I1{});
    }
  };

  using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>; using T3 = std::integral_constant<int, 3>; using T4 = std::integral_constant<int, 4>;
  int nchunk = 0;
  while (true) {
    const int tn_rt = cur.tn;
    if (tn_rt >= 3) { kloop(T3{}); retire(T3{}, cur); }
    else if (tn_rt == 2) { kloop(T2{}); retire(T2{}, cur); }
    else { kloop(T1{}); retire(T1{}, cur); }
    ++nchunk;
    if (nxt.nk == 0) break;
    cur = nxt;
    nxt = next_chunk(a, wk);
    lane_offsets(nxt, va_n, vb_n, Lk_n, Lkend_n, LzA_n, LzB_n);
  }
  // the last chunk's epilogue has nothing to hide under
  piece(I0{}, I0{}); piece(I0{}, I1{}); piece(I1{}, I0{}); piece(I1{}, I1{}); piece(I2{}, I0{}); piece(I2{}, I1{});
  if (TRACE && threadIdx.x == 0) { tr[3] = __builtin_amdgcn_s_memtime(); tr[4] = wall_clock64(); tr[5] = nchunk; for (int i = 0; i < 5; ++i) tr[8 + i] = trc[i]; }
}

template <bool TRACE>
__global__ __launch_bounds__(512, 2) void gemm_pk3_kernel(PkArgs a) {
  constexpr int NS = 4, PANEL = 16 * 128, STAGE = 2 * PANEL;
  __shared__ __attribute__((aligned(16))) float smem[2 * NS * STAGE + 8 * 16 * 36];
  const int g = blockIdx.x, G = gridDim.x;
  // eight waves = two TEAMS of four (waves 0-3, 4-7; one wave of each per SIMD).  A team is what a workgroup of version 2 was: it
  // has its own run of units, its own LDS ring and accumulators.  What is new is that the teams alternate by construction: while one
  // team's waves are in the MFMA block of a slice, the other's are in the non-MFMA block of theirs (DMA issues, write-behind piece,
  // fragment reads), the two workgroup-wide barriers per slice being the hand-over.  Two independent workgroups per CU lock IN phase
  // instead (both in the non-MFMA block, then both sharing the matrix pipe: gemm_lab traces, 67-85 % of the MFMA floor).
  const int wfull = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), team = wfull >> 2;
  auto walker_of = [&](int tm) __attribute__((always_inline)) {
    Walker k;
    const int x = g & 7, j = (g >> 3) * 2 + tm, J = (G >> 3) * 2;
    const int CT = a.Z * a.UN;
    k.cx0 = (int)((long long)x * CT / 8);
    k.nx = (int)((long long)(x + 1) * CT / 8) - k.cx0;
    const long long T = (long long)a.S * a.tilesM * k.nx;
    k.q = k.nx > 0 ? (long long)j * T / J : 0;
    k.q1 = k.nx > 0 ? (long long)(j + 1) * T / J : 0;
    return k;
  };
  auto slices_of = [&](int tm) __attribute__((always_inline)) {
    Walker k = walker_of(tm);
    int n = 0;
    while (true) { const Chunk c = next_chunk(a, k); if (c.nk == 0) break; n += c.nk; }
    return n;
  };
  const int bars_mine = 1 + team + 2 * slices_of(team), bars_other = 2 - team + 2 * slices_of(1 - team);     // barriers each team executes
  Walker wk = walker_of(team);
  unsigned long long* tr = nullptr;
  if (TRACE) {
    tr = a.trace + (size_t)g * 64;
    if (threadIdx.x == 0) { tr[0] = __builtin_amdgcn_s_memtime(); tr[1] = wall_clock64(); }
  }
  Chunk cur = next_chunk(a, wk);
  Chunk nxt = next_chunk(a, wk);

  const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
  const int w = wfull & 3;                            // wave within the team
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem + (unsigned)(team * NS * STAGE) * 4u;
  const bool partial = a.S > 1;
  const long long nY = (long long)(a.Z - 1) * a.zy + (long long)a.B * a.M * a.HW;            // elements of one output (partial: one slice)
  const i32x4 ar = make_desc(a.at, (int)(((long long)(a.Z - 1) * a.zat + (long long)a.K * a.lda) * 4));
  const i32x4 xr = make_desc(a.x, (int)(((long long)(a.Z - 1) * a.zx + (long long)a.K * a.N) * 4));
  const __amdgpu_buffer_rsrc_t yr = make_rsrc(partial ? a.ws : a.y, (int)((partial ? a.slice * a.S : nY) * 4));
  const bool has_add = !partial && a.addend != nullptr;
  const __amdgpu_buffer_rsrc_t addr = make_rsrc(has_add ? a.addend : a.y, has_add ? (int)(nY * 4) : 0);
  const __amdgpu_buffer_rsrc_t br = make_rsrc(a.bias ? a.bias : a.y, (!partial && a.bias) ? a.M * 4 : 0);
  const float lowest = (!partial && a.epi == PRN_EPI_RELU) ? 0.f : -__builtin_inff();     // ReLU or nothing (sigmoid epilogues stay on version 1)

  // ---- load side.  BRANCH-FREE by construction: four LDS-DMA issues every slice, for ever; when the stream crosses into the next
  // chunk the lane offsets / scalar offsets are swapped in by selects (the next chunk's are prepared when that chunk becomes known),
  // and behind the last chunk they are OOB, i.e. the issues become no-ops that move nothing -- so the number of loads in flight is a
  // constant and s_waitcnt vmcnt takes an immediate.  (tools/native/issue_cost.cpp: an LDS-DMA, a buffer load, a ds_write or an
  // s_barrier between two MFMAs costs the wave nothing; what cost versions 2 / 3 of this kernel 10-25 % were TAKEN BRANCHES in the
  // slice loop -- a wave refetches its instruction stream after each, and the loop had ~10 of them per slice.)
  // This lane's pieces of the two panels are rows 4w + 2i + hi (i = 0, 1), columns 4*lo .. 4*lo + 3.
  const int c4 = lo * 4;
  int ka[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) ka[i] = a.K - (4 * w + 2 * i + hi);
  unsigned va[2], vb[2], va_n[2], vb_n[2];
  int Lk, Lkend, LzA, LzB, Lk_n, Lkend_n, LzA_n, LzB_n, ls = 0;
  auto lane_offsets = [&](const Chunk& c, unsigned (&oa)[2], unsigned (&ob)[2], int& k, int& kend, int& zA, int& zB) __attribute__((always_inline)) {
    const bool live = c.nk != 0;
    const int n = c.n0 + c4;
    const bool bok = live && c4 < 32 * c.tn && n < a.N;
    const int b = bok ? n / a.HW : 0, p = n - b * a.HW;
    const bool aok = live && c.m0 + c4 < a.M;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 4 * w + 2 * i + hi;
      oa[i] = aok ? (unsigned)(r * a.lda + c.m0 + c4) * 4u : OOB;
      ob[i] = bok ? (unsigned)((b * a.K + r) * a.HW + p) * 4u : OOB;
    }
    k = c.kt0; kend = live ? c.kt0 + c.nk : 0x7fffffff;           // (behind the last chunk: never crosses again)
    zA = (int)(c.z * a.zat * 4); zB = (int)(c.z * a.zx * 4);
  };
  auto issue_slice = [&]() __attribute__((always_inline)) {
    const int k0 = Lk * 16;
    const unsigned st = lds0 + (unsigned)(ls * STAGE + 4 * w * 128) * 4u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool kok = k0 < ka[i];
      lds_dma16(st + i * 1024, ar, kok ? va[i] : OOB, LzA + k0 * a.lda * 4);
      lds_dma16(st + PANEL * 4 + i * 1024, xr, kok ? vb[i] : OOB, LzB + k0 * a.HW * 4);
    }
    ls = (ls + 1) & (NS - 1);
    ++Lk;
    const bool cross = Lk == Lkend;                    // selects, not a branch
#pragma unroll
    for (int i = 0; i < 2; ++i) { va[i] = cross ? va_n[i] : va[i]; vb[i] = cross ? vb_n[i] : vb[i]; }
    Lk = cross ? Lk_n : Lk; LzA = cross ? LzA_n : LzA; LzB = cross ? LzB_n : LzB;
    Lkend = cross ? Lkend_n : Lkend;
  };

  // ---- compute side
  f32x16 acc[3], accP[3];
  const float* abase = smem + team * NS * STAGE + hi * 128 + 32 * w + lo;
  const float* bbase = smem + team * NS * STAGE + PANEL + hi * 128 + lo;
  int cs = 0;                                         // stage of the slice being computed
  float* pad = smem + 2 * NS * STAGE + wfull * (16 * 36);
  const int crow = lane >> 3, ccol = (lane & 7) * 4;
  unsigned pbase[3] = {OOB, OOB, OOB};           // write-behind state of the previous chunk; OOB: nothing to store
  int mrowP = 0;
  float bm[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accP[j][r] = 0.f;

  lane_offsets(cur, va, vb, Lk, Lkend, LzA, LzB);
  lane_offsets(nxt, va_n, vb_n, Lk_n, Lkend_n, LzA_n, LzB_n);
  issue_slice(); issue_slice(); issue_slice();
  wait_vm<8>();
  __builtin_amdgcn_s_barrier();                        // both teams: slice 0 is in LDS
  if (team) __builtin_amdgcn_s_barrier();              // team 1 runs half a slice behind team 0

#define SB __builtin_amdgcn_sched_barrier(0)
  // Write-behind epilogue of the previous chunk, piece (j, h) = rows 16h .. 16h + 15 of column block j: eight accumulator registers
  // through the wave's 16 x 36 LDS pad come back as two float4 rows -> bias / addend / ReLU -> 16-byte stores, 128 contiguous bytes
  // per 8 lanes.  Always all six pieces: column blocks the previous chunk did not have carry the OOB offset and store nothing.
  auto piece = [&](auto jt, auto ht) __attribute__((always_inline)) {
    constexpr int j = decltype(jt)::value, h = decltype(ht)::value;
#pragma unroll
    for (int r = 0; r < 8; ++r) pad[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36 + lo] = accP[j][8 * h + r];
    __builtin_amdgcn_wave_barrier();
    f32x4 v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) v[q] = *reinterpret_cast<const f32x4*>(&pad[(crow + 8 * q) * 36 + ccol]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned off = (mrowP + 16 * h + 8 * q < a.M) ? pbase[j] + (unsigned)((16 * h + 8 * q) * a.HW) * 4u : OOB;
      f32x4 o = v[q];
      if (has_add) o += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(addr, (int)off, 0, 0));
      o += bm[h][q];
      o.x = fmaxf(o.x, lowest); o.y = fmaxf(o.y, lowest); o.z = fmaxf(o.z, lowest); o.w = fmaxf(o.w, lowest);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yr, (int)off, 0, 0);
    }
  };
  // the finished chunk becomes the "previous" one: accumulators to the second set, output offsets and bias of its rows
  auto retire = [&](auto tn, const Chunk& c) __attribute__((always_inline)) {
    constexpr int TN = decltype(tn)::value;
    mrowP = c.m0 + 32 * w + crow;
    const long long zoff = (long long)c.z * a.zy + (partial ? (long long)c.s * a.slice : 0);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j < TN) {
        accP[j] = acc[j];
        const int n = c.n0 + 32 * j + ccol;
        const bool nok = n < a.N;
        const int b = nok ? n / a.HW : 0, p = n - b * a.HW;
        pbase[j] = nok ? (unsigned)(zoff + (long long)(b * a.M + mrowP) * a.HW + p) * 4u : OOB;
      } else {
        pbase[j] = OOB;
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 2; ++q) bm[h][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, (mrowP + 16 * h + 8 * q) * 4, 0, 0));
  };

  // One slice of a team: NON-MFMA block [four LDS-DMA issues for the slice three ahead] [optional write-behind piece] [all fragment
  // reads] | barrier | MFMA block [8 * TN MFMAs back to back] [slice + 1 landed: vmcnt] | barrier.  The other team is in the other
  // block.  YOUNG = epilogue stores issued since the group being waited for was (two load groups of four always stay in flight).
  auto slice = [&](auto tn, auto young, auto pj, auto ph) __attribute__((always_inline)) {
    constexpr int TN = decltype(tn)::value, J = decltype(pj)::value;
    issue_slice();
    if constexpr (J >= 0) piece(pj, ph);
    const float* as = abase + cs * STAGE;
    const float* bs = bbase + cs * STAGE;
    float av[8], bv[8][TN];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      av[kk] = as[kk * 256];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[kk][j] = bs[kk * 256 + 32 * j];
    }
    SB;
    __builtin_amdgcn_s_barrier();
    SB;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk][j], acc[j], 0, 0, 0);
    SB;
    wait_vm<8 + decltype(young)::value>();
    __builtin_amdgcn_s_barrier();
    cs = (cs + 1) & (NS - 1);
  };
  using N1 = std::integral_constant<int, -1>;
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  using Y0 = std::integral_constant<int, 0>; using Y2 = std::integral_constant<int, 2>; using Y4 = std::integral_constant<int, 4>; using Y6 = std::integral_constant<int, 6>;
  // all slices of the current chunk: the first six carry the six pieces of the previous chunk (straight-line code, the piece of
  // each slice is a compile-time one), the rest is the bare loop.  nk >= 4 (host); chunks shorter than six slices flush the
  // remaining pieces behind their last slice.
  auto kloop = [&](auto tn) __attribute__((always_inline)) {
    constexpr int TN = decltype(tn)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int nk = cur.nk;
    slice(tn, Y2{}, I0{}, I0{});
    slice(tn, Y4{}, I0{}, I1{});
    slice(tn, Y6{}, I1{}, I0{});
    slice(tn, Y6{}, I1{}, I1{});
    if (__builtin_expect(nk >= 6, 1)) {
      slice(tn, Y6{}, I2{}, I0{});
      slice(tn, Y6{}, I2{}, I1{});
#pragma nounroll
      for (int t = 6; t < nk; ++t) slice(tn, Y0{}, N1{}, N1{});
    } else {
#pragma nounroll
      for (int t = 4; t < nk; ++t) slice(tn, Y0{}, N1{}, N1{});
      piece(I2{}, I0{}); piece(I2{}, I1{});
    }
  };

  using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>; using T3 = std::integral_constant<int, 3>; using T4 = std::integral_constant<int, 4>;
  int nchunk = 0;
  while (cur.nk != 0) {
    const int tn_rt = cur.tn;
    if (tn_rt >= 3) { kloop(T3{}); retire(T3{}, cur); }
    else if (tn_rt == 2) { kloop(T2{}); retire(T2{}, cur); }
    else { kloop(T1{}); retire(T1{}, cur); }
    ++nchunk;
    if (nxt.nk == 0) break;
    cur = nxt;
    nxt = next_chunk(a, wk);
    lane_offsets(nxt, va_n, vb_n, Lk_n, Lkend_n, LzA_n, LzB_n);
  }
  for (int i = bars_mine; i < (bars_mine > bars_other ? bars_mine : bars_other); ++i) __builtin_amdgcn_s_barrier();     // the other team is still at work
  // the last chunk's epilogue has nothing to hide under
  piece(I0{}, I0{}); piece(I0{}, I1{}); piece(I1{}, I0{}); piece(I1{}, I1{}); piece(I2{}, I0{}); piece(I2{}, I1{});
  if (TRACE && threadIdx.x == 0) { tr[3] = __builtin_amdgcn_s_memtime(); tr[4] = wall_clock64(); tr[5] = nchunk; }
}

// K splits: enough units for an even deal (>= ~4 per workgroup), each split at least 4 slices deep.
int plan_splits(long long units, int KT, int G) {
  static int forced = -1;
  if (forced < 0) { const char* e = getenv("PRN_PK_SPLITS"); forced = e ? atoi(e) : 0; }
  if (forced > 0) return forced <= KT ? forced : KT;
  int best = 1;
  double best_cost = 1e30;
  for (int s = 1; s <= 16 && s * 4 <= KT; ++s) {
    const long long u = units * s;
    const long long per = (u + G - 1) / G;                       // units of the most loaded workgroup
    const double depth = (double)KT / s + 3.0;                   // slices per unit + prologue / epilogue, in slice times
    const double cost = per * depth * (s > 1 ? 1.03 : 1.0);      // partial slices cost a little traffic and the sum kernel
    if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
  }
  return best;
}

int pk_grid() {
  static int g = -1;
  if (g < 0) { const char* e = getenv("PRN_PK_GRID"); g = e ? atoi(e) : 512; if (g < 8) g = 8; g &= ~7; }
  return g;
}

}  // namespace

extern "C" int64_t prn_gemm_kn_ws_bytes(const prn_gemm_desc* d) {
  if (!d || d->M <= 0 || d->K <= 0 || d->B <= 0 || d->HW <= 0 || d->nz <= 0) return -1;
  const long long N = (long long)d->B * d->HW;
  const long long units = (long long)cdiv(d->M, 128) * cdiv(N, 32) * d->nz;
  const int S = plan_splits(units, cdiv(d->K, 16), pk_grid());
  return S > 1 ? (int64_t)S * d->nz * N * d->M * 4 : 0;
}

extern "C" int prn_gemm_kn(const prn_gemm_desc* d, const float* at, const float* x, const float* bias, const float* addend, float* y, void* ws,
                           void* stream) {
  PRN_REQUIRE(d && at && x && y, "prn_gemm_kn: null argument");
  PRN_REQUIRE(d->M > 0 && d->K > 0 && d->B > 0 && d->HW > 0 && d->nz > 0, "prn_gemm_kn: bad sizes");
  PRN_REQUIRE((d->HW & 3) == 0 && (d->M & 3) == 0 && (d->lda & 3) == 0 && d->lda >= d->M, "prn_gemm_kn: HW, M, lda must be multiples of 4 (HW=%d M=%d lda=%d)",
              d->HW, d->M, d->lda);
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  PRN_REQUIRE(al16(at) && al16(x) && al16(y) && al16(addend) && al16(ws), "prn_gemm_kn: operands must be 16-byte aligned");
  PRN_REQUIRE((d->zat & 3) == 0 && (d->zx & 3) == 0 && (d->zy & 3) == 0, "prn_gemm_kn: batch strides must be multiples of 4");
  const long long N = (long long)d->B * d->HW;
  PRN_REQUIRE((long long)d->K * N < (1LL << 29) && (long long)d->K * d->lda < (1LL << 29) && N < (1LL << 30), "prn_gemm_kn: operand larger than a buffer descriptor");
  PkArgs a;
  a.at = at; a.x = x; a.bias = bias; a.addend = addend; a.y = y; a.ws = (float*)ws;
  a.M = d->M; a.K = d->K; a.B = d->B; a.HW = d->HW; a.N = (int)N; a.lda = d->lda; a.Z = d->nz; a.epi = d->epilogue;
  a.KT = cdiv(d->K, 16);
  a.UN = cdiv(N, 32);
  a.tilesM = cdiv(d->M, 128);
  a.xbytes = (int)((long long)d->K * N * 4);
  a.abytes = (int)((long long)d->K * d->lda * 4);
  a.zat = d->zat; a.zx = d->zx; a.zy = d->zy;
  a.slice = (long long)d->nz * N * d->M;
  const int G = pk_grid();
  a.S = plan_splits((long long)a.tilesM * a.UN * a.Z, a.KT, G);
  if (a.S > 1) {
    PRN_REQUIRE(ws != nullptr, "prn_gemm_kn: workspace required (K split %d)", a.S);
    PRN_REQUIRE(d->nz == 1 || d->zy == (long long)N * d->M, "prn_gemm_kn: a K split needs densely packed outputs");
  }
  static int ns = -1;
  if (ns < 0) { const char* e = getenv("PRN_PK_STAGES"); ns = e ? atoi(e) : 4; }
  hipStream_t st = (hipStream_t)stream;
  a.trace = nullptr;
  a.dbg = getenv("PRN_PK_DEBUG") ? atoi(getenv("PRN_PK_DEBUG")) : 0;
  static int ver = -1;
  if (ver < 0) { const char* e = getenv("PRN_PK_V"); ver = e ? atoi(e) : 2; }
  if (ver >= 2 && a.KT / a.S >= 4 && a.epi != PRN_EPI_SIGMOID) {
    const long long nY = (long long)(a.Z - 1) * a.zy + (long long)a.B * a.M * a.HW;
    PRN_REQUIRE((a.S > 1 ? a.slice * a.S : nY) < (1LL << 29) && (long long)(a.Z - 1) * a.zat + (long long)a.K * a.lda < (1LL << 29) &&
                    (long long)(a.Z - 1) * a.zx + (long long)a.K * a.N < (1LL << 29),
                "prn_gemm_kn: tensor larger than a buffer descriptor");
    if (ver == 3) {
      hipLaunchKernelGGL((gemm_pk3_kernel<false>), dim3(G / 2), dim3(512), 0, (hipStream_t)stream, a);
    } else if (const char* e = getenv("PRN_PK_TRACE")) {
      a.trace = (unsigned long long*)strtoull(e, nullptr, 0);
      hipLaunchKernelGGL((gemm_pk2_kernel<true>), dim3(G), dim3(256), 0, (hipStream_t)stream, a);
    } else {
      hipLaunchKernelGGL((gemm_pk2_kernel<false>), dim3(G), dim3(256), 0, (hipStream_t)stream, a);
    }
    PRN_CHECK_LAUNCH("prn_gemm_kn");
    if (a.S > 1) return prn_launch_reduce_epilogue(a.ws, bias, addend, y, a.slice, d->M, d->HW, a.S, d->epilogue, (hipStream_t)stream);
    return 0;
  }
  if (const char* e = getenv("PRN_PK_TRACE")) {          // lab: address of a device buffer of gridDim.x * 64 uint64
    a.trace = (unsigned long long*)strtoull(e, nullptr, 0);
    hipLaunchKernelGGL((gemm_pk_kernel<4, true>), dim3(G), dim3(256), 0, st, a);
  } else if (ns == 3) hipLaunchKernelGGL((gemm_pk_kernel<3>), dim3(G), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gemm_pk_kernel<4>), dim3(G), dim3(256), 0, st, a);
  PRN_CHECK_LAUNCH("prn_gemm_kn");
  if (a.S > 1) return prn_launch_reduce_epilogue(a.ws, bias, addend, y, a.slice, d->M, d->HW, a.S, d->epilogue, st);
  return 0;
}

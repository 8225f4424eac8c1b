// Convolution as implicit GEMM on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
//
//   forward / dgrad :  Y[M x N] = W[M x K] * im2col(X)[K x N]      M = Cout, N = B*Ho*Wo, K = Cin*KH*KW
//   wgrad           : dW[M x K] = dY[M x N] * im2col(X)^T[N x K]   (reduction over pixels, deterministic split)
//
// NCHW makes the pixel index the contiguous one, so both im2col(X) rows and Y rows are streamed with
// coalesced 256-byte wave accesses; padding / reflection / nearest-x2 / zero-dilation are folded into the
// operand gather (no padded or upsampled tensor ever exists in HBM).  A 256-thread workgroup (4 waves, 2x2)
// owns a (64*TM)x(64*TN) tile; each wave accumulates TMxTN 32x32 MFMA tiles in registers.  Operands are staged
// through LDS in 16-deep K slices, double buffered, with the next slice's global loads in flight during the
// MFMA loop.  LDS rows that are read "down a column" by the MFMA operand pattern use a 17-float pitch, which
// makes the 32-lane-group reads conflict-free.
#include <stddef.h>
#include <stdlib.h>
#include <type_traits>
#include "prn_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

// Ragged batches: up to 6 segments, each a dense [B, C, H_s, W_s] tensor, stored back to back (the five SOLO grid levels
// run through the same weights).  The GEMM's pixel axis is the concatenation of all segments' pixels; a tile / pixel chunk
// never straddles two segments (host-side alignment check), so a workgroup looks its segment up once (scalar code) and then
// works on plain dense geometry.
constexpr int MAX_SEG = 6;
struct Seg {
  int nseg;                 // 0: dense tensor
  int n0[MAX_SEG + 1];      // first pixel of segment s on the GEMM's pixel axis; n0[nseg] = total
  int H[MAX_SEG], W[MAX_SEG];
  int x0[MAX_SEG], y0[MAX_SEG];   // element offsets of the segment inside the packed input / output tensor
};
__device__ __forceinline__ int seg_find(const Seg& g, int n) {
  int s = 0;
#pragma unroll
  for (int t = 1; t < MAX_SEG; ++t)
    if (t < g.nseg && n >= g.n0[t]) s = t;
  return s;
}
// Entry `s` of one of the table's arrays, read straight from the kernel-argument segment with a scalar load (the table is
// the last member of the single by-value kernel argument).  Indexing the by-value copy dynamically, or overwriting fields
// of it, made hipcc spill the whole argument struct to scratch and wrap the gathers in waterfall loops (84 -> 117 ms/step).
template <class Args>
__device__ __forceinline__ int seg_read(size_t member_offset, int s) {
  typedef __attribute__((address_space(4))) const int* kptr;
  kptr base = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
  return base[(offsetof(Args, seg) + member_offset) / 4 + s];
}
#define SEG_READ(Args, member, s) seg_read<Args>(offsetof(Seg, member), s)

struct ConvArgs {
  const float* x; const float* w; const float* bias; const float* addend; float* y;
  int B, C, H, W, M, stride, pad, Ho, Wo, epi;
  int K, N, HoWo, HW, tilesM, nblocks, splits;
  int xbytes, wbytes;
  int ystride, yW, yHW;     // output map: GEMM pixel (oh, ow) is stored at (oh*ystride + phase_y, ow*ystride + phase_x) of a yHW plane
  int zx, zw, zy;           // VEC instances: element strides of x / w / y per blockIdx.z (prn_gemm_batched; 0 for a plain conv)
  int wide_store;           // epilogue through the LDS transpose (float4 stores): output / addend / workspace 16-byte aligned
  int ilv;                  // K loop with the staging work issued inside the MFMA loop (see conv_igemm_kernel)
  int tail_first, tail_splits;   // tail split (see plan_tail): blocks >= tail_first are K-split pieces of the last tiles; 0 splits = off
  float* ws;
  unsigned* cnt;                 // per-tile arrival counters (zero between launches) when the K-split sum is folded into this kernel, else null
  Seg seg;
};

// Operand fetches go through buffer descriptors: a lane whose element does not exist (padding, tile tails) carries the
// byte offset OOB, which the hardware range check turns into a 0.0 result without touching memory.  No select depends
// on the loaded value, so the loads of slice k+1 stay in flight across the whole MFMA loop of slice k and are first
// waited for at the LDS stores (with value-side selects hipcc put s_waitcnt vmcnt(0) in front of the MFMAs).
constexpr unsigned OOB = 0x80000000u;
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
}
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  // (bit_cast the whole vector: __builtin_bit_cast on a single vector element reads element 0 with hipcc 7.2)
  const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
  return make_float4(q.x, q.y, q.z, q.w);
}

// Agent-scope accesses (sc1): the store is written through to memory, the load is served from memory -- visible across the
// eight XCDs, whose L2s are not coherent with each other, without a cache-wide write-back / invalidate.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bstore4_agent(__amdgpu_buffer_rsrc_t r, unsigned voff, float4 v) {
  const f32x4 q = {v.x, v.y, v.z, v.w};
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q), r, (int)voff, 0, 16);
}
__device__ __forceinline__ float4 bload4_agent(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 16));
  return make_float4(q.x, q.y, q.z, q.w);
}

__device__ __forceinline__ int reflect_idx(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * n - 2 - i : i;
}

// Offset (inside one channel plane) of the input sample that virtual im2col position (ih, iw) reads, or -1 when that
// position contributes zero.  All loads built on it are UNCONDITIONAL: a per-lane branch around a load makes hipcc
// serialise the gather behind s_waitcnt, which left the first version of this kernel latency-bound at ~30 % of the
// MFMA rate.
template <int MODE>
__device__ __forceinline__ int tap_offset(int ih, int iw, bool ok, int H, int W) {
  if (MODE == PRN_IN_ZERO) {
    ok = ok && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
  } else if (MODE == PRN_IN_REFLECT) {
    ih = reflect_idx(ih, H);
    iw = reflect_idx(iw, W);
  } else if (MODE == PRN_IN_UP2_REFLECT) {
    ih = reflect_idx(ih, 2 * H) >> 1;
    iw = reflect_idx(iw, 2 * W) >> 1;
  } else if (MODE == PRN_IN_UP2_PHASE) {   // replicate border (one sub-pixel phase of nearest-x2 + reflect-pad + 3x3)
    ih = ih < 0 ? 0 : (ih >= H ? H - 1 : ih);
    iw = iw < 0 ? 0 : (iw >= W ? W - 1 : iw);
  } else {  // PRN_IN_DILATED (factor 2): only even virtual coordinates carry data
    ok = ok && ih >= 0 && iw >= 0 && ((ih | iw) & 1) == 0;
    ih >>= 1;
    iw >>= 1;
    ok = ok && ih < H && iw < W;
  }
  return ok ? ih * W + iw : -1;
}

// BK = K-slice depth per staging round.  Small-N layers (backbone stages 3-4: N = 9600 / 2400 pixels) cannot fill 256 CUs
// with output tiles alone, so the launcher also splits K across blockIdx.y (deterministic: partial tiles go to a
// workspace that reduce_epilogue_kernel sums in fixed order before bias / addend / activation).
// VEC (1x1, stride 1, no padding, H*W % 4 == 0): the im2col operand is the activation matrix itself, staged with float4
// loads along the pixel axis and 128-bit LDS stores.
// WM = waves along M (2: the 2x2 wave grid; 1: all four waves side by side along the pixels, a 32 x 128 tile for layers
// with <= 32 output channels -- the 27-channel offset/modulator conv of every DCN block wasted 58 % of a 64-row tile).
// V3 (3x3, stride 1, zero padding 1, W % 4 == 0): the im2col operand is fetched as float4 too -- four consecutive output
// pixels of one row read four consecutive input pixels for every tap; at the left / right image border the vector is
// loaded one pixel further in and shifted (with a zero) when it is stored to LDS.  4x fewer gather instructions.
template <int KS, int MODE, int TM, int TN, int BK, bool VEC = false, int WM = 2, int WN = 4 / WM, bool V3 = false>
__global__ __launch_bounds__(64 * WM * WN, (TM * TN == 4 ? 4 : 1)) void conv_igemm_kernel(ConvArgs a) {   // 128x128: cap at 128 VGPRs -> 4 waves/SIMD (+7..15 %)
  static_assert(!VEC || (KS == 1 && MODE == PRN_IN_ZERO), "vector staging is the plain-GEMM case");
  static_assert(!V3 || (KS == 3 && MODE == PRN_IN_ZERO && !VEC), "V3 is the 3x3 zero-padded stride-1 case");
  constexpr int NT = 64 * WM * WN;               // threads per workgroup
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, LDA = BK + 1;
  constexpr int KSTEP = NT / BN;   // K rows covered by one sweep of the block
  constexpr int NB = BK / KSTEP;   // gathered elements per thread per K slice
  constexpr int AQ = BK / 4;       // float4 groups per A row
  constexpr int AROWS = NT / AQ;   // A rows covered by one sweep
  constexpr int NA = BM >= AROWS ? BM / AROWS : 1;   // float4 loads per thread per K slice
  constexpr bool AALL = BM >= AROWS;                 // every thread stages A (otherwise only those with arow < BM)
  constexpr int KK = KS * KS;
  constexpr int VG = BN / 4;       // VEC: float4 pixel groups per K row
  constexpr int VROWS = NT / VG;   // VEC: K rows per sweep
  constexpr int NBV = BK / VROWS;  // VEC: float4 loads per thread per K slice
  // one LDS allocation: the two operand double buffers during the K loop, then (epilogue) one 32 x 36 transpose pad per wave
  constexpr int AS = 2 * BM * LDA, BS = 2 * BK * BN, CPITCH = 36, CS = (NT / 64) * 32 * CPITCH;
  static_assert(AS % 4 == 0, "the im2col buffers must stay 16-byte aligned");
  __shared__ __attribute__((aligned(16))) float smem[(AS + BS) > CS ? (AS + BS) : CS];
  float (*As)[BM * LDA] = reinterpret_cast<float (*)[BM * LDA]>(smem);
  float (*Bs)[BK * BN] = reinterpret_cast<float (*)[BK * BN]>(smem + AS);
  __shared__ unsigned taps[KS > 1 ? KK * BN : 1];   // per-pixel tap byte offsets (or OOB), built once per workgroup
                                                    // (V3: per 4-pixel group: offset of the vector | border code in bits 0-1)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // block -> (tile, K split).  Tail split: a launch of R full residency rounds plus a fraction leaves the CUs mostly idle
  // through the last, partly filled round; the tiles of that round are therefore cut along K into `tail_splits` short
  // workgroups each (blocks >= tail_first), which fill the machine as the full-length workgroups retire.  Their partial
  // tiles go to the workspace and reduce_tail_kernel sums them in fixed order.
  int id, ksplit = blockIdx.y, nsplit = a.splits;
  if (a.tail_splits > 1 && (int)blockIdx.x >= a.tail_first) {
    const int q = (int)blockIdx.x - a.tail_first;
    id = a.tail_first + q / a.tail_splits;
    ksplit = q - (q / a.tail_splits) * a.tail_splits;
    nsplit = a.tail_splits;
  } else {
    id = prn_xcd_remap(blockIdx.x, a.tail_splits > 1 ? a.tail_first : a.nblocks);
  }
  const int m0 = (id % a.tilesM) * BM;
  int n0 = (id / a.tilesM) * BN;
  // geometry of the tensor this workgroup works on: the descriptor's, or its segment's in a ragged batch
  int H_ = a.H, W_ = a.W, HW_ = a.HW, Ho_ = a.Ho, Wo_ = a.Wo, HoWo_ = a.HoWo, N_ = a.N, xbytes_ = a.xbytes;
  const float* x_ = a.x;
  const float* add_ = a.addend;
  float* y_ = a.y;
  if (a.seg.nseg > 0) {
    const int sg = seg_find(a.seg, n0);
    H_ = Ho_ = SEG_READ(ConvArgs, H, sg);
    W_ = Wo_ = SEG_READ(ConvArgs, W, sg);
    HW_ = HoWo_ = H_ * W_;
    const int yo = SEG_READ(ConvArgs, y0, sg);
    x_ += SEG_READ(ConvArgs, x0, sg);
    y_ += yo;
    if (add_) add_ += yo;
    xbytes_ = a.B * a.C * HW_ * 4;
    N_ = a.B * HoWo_;
    n0 -= SEG_READ(ConvArgs, n0, sg);
  }
  (void)Ho_;

  // PRN_IN_UP2_PHASE: blockIdx.z = output phase (py, px); each phase has its own [M x 4C] weight matrix, reads the 2x2
  // source window starting at (i + py - 1, j + px - 1) and stores to (2i + py, 2j + px)
  const int py = (MODE == PRN_IN_UP2_PHASE) ? (int)(blockIdx.z >> 1) : 0, px = (MODE == PRN_IN_UP2_PHASE) ? (int)(blockIdx.z & 1) : 0;
  const int pad_y = (MODE == PRN_IN_UP2_PHASE) ? 1 - py : a.pad, pad_x = (MODE == PRN_IN_UP2_PHASE) ? 1 - px : a.pad;
  const float* wz = (MODE == PRN_IN_UP2_PHASE) ? a.w + (size_t)blockIdx.z * a.M * a.K : a.w;
  if (VEC) {                                             // batched GEMM: blockIdx.z selects the (weights, activations, output) triple
    x_ += (size_t)blockIdx.z * a.zx;
    wz += (size_t)blockIdx.z * a.zw;
    y_ += (size_t)blockIdx.z * a.zy;
  }
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(x_, xbytes_), wr = make_rsrc(wz, a.wbytes);

  // pixel owned by this thread in the B (im2col) operand; its K rows are wave-uniform
  const int nl = tid % BN;
  // K row of this thread inside a sweep: wave-uniform (-> SGPR address math) when a wave spans one row of the tile
  const int krow0 = (BN >= 64) ? __builtin_amdgcn_readfirstlane(tid / BN) : tid / BN;
  const int n = n0 + nl;
  const bool nvalid = n < N_;
  int b = 0, oh = 0, ow = 0;
  if (nvalid) {
    b = n / HoWo_;
    const int p = n - b * HoWo_;
    oh = p / Wo_;
    ow = p - oh * Wo_;
  }
  const int ih0 = oh * a.stride - pad_y, iw0 = ow * a.stride - pad_x;
  const int pix0 = b * a.C * HW_;                      // element offset of this pixel's image
  unsigned off1 = OOB;                                   // byte offsets (or OOB) relative to x_, channel 0
  if (KS == 1) {
    const int o = tap_offset<MODE>(ih0, iw0, nvalid, H_, W_);
    off1 = o >= 0 ? (unsigned)(pix0 + o) * 4u : OOB;
  } else if (V3) {
    for (int idx = tid; idx < KK * VG; idx += NT) {
      const int rs = idx / VG, g = idx - rs * VG;
      const int r = rs / 3, sx = rs - r * 3;
      const int n4 = n0 + g * 4;                           // first of four pixels in one output row (Wo % 4 == 0)
      unsigned e = OOB;
      if (n4 < N_) {
        const int b4 = n4 / HoWo_, p4 = n4 - b4 * HoWo_, oh4 = p4 / Wo_, ow4 = p4 - oh4 * Wo_;
        const int ih = oh4 + r - 1, iw = ow4 + sx - 1;
        if ((unsigned)ih < (unsigned)H_) {
          const unsigned code = iw < 0 ? 1u : (iw + 3 >= W_ ? 2u : 0u);      // 1: loaded one pixel to the right, 2: one to the left
          const int iwl = iw < 0 ? iw + 1 : (iw + 3 >= W_ ? iw - 1 : iw);
          e = ((unsigned)(b4 * a.C * HW_ + ih * W_ + iwl) * 4u) | code;
        }
      }
      taps[idx] = e;
    }
    __syncthreads();
  } else {
    for (int t = krow0; t < KK; t += KSTEP) {
      const int r = t / KS, s = t - r * KS;
      const int o = tap_offset<MODE>(ih0 + r, iw0 + s, nvalid, H_, W_);
      taps[t * BN + nl] = o >= 0 ? (unsigned)(pix0 + o) * 4u : OOB;
    }
    __syncthreads();
  }

  // A (weight) operand: NA float4 groups per thread; rows beyond M fall outside the descriptor by themselves
  const int arow = tid / AQ, akq = (tid % AQ) * 4;
  const bool k4 = ((a.K & 3) == 0) && ((reinterpret_cast<uintptr_t>(wz) & 15) == 0);
  unsigned abase[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int m = m0 + arow + AROWS * i;
    abase[i] = (m < a.M && (AALL || arow < BM)) ? (unsigned)(m * a.K + akq) * 4u : OOB;
  }

  // VEC staging: this thread's group of 4 consecutive pixels, K row vrow0 (+ VROWS * i)
  const int vg = tid % VG, vrow0 = tid / VG;
  unsigned vbase = OOB;
  if (VEC) {
    const int nv = n0 + vg * 4;
    if (nv < N_) {
      const int bv = nv / HoWo_;
      vbase = (unsigned)((bv * a.C + vrow0) * HW_ + (nv - bv * HoWo_)) * 4u;
    }
  }

  float ra[NA][4];
  float rb[(VEC || V3) ? 1 : NB];
  float4 rv[(VEC || V3) ? NBV : 1];
  unsigned rcode = 0;                                      // V3: 2-bit border codes of this slice's vectors
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // K4 (weight rows readable as float4) is a compile-time tag and the last slice is peeled, so the steady-state loop
  // body is branch-free: with a uniform branch around the loads, or "if (kt + 1 < KT)" around load and store, hipcc's
  // waitcnt insertion has to assume loads left pending by the not-taken path and drains vmcnt before the next gather.
  auto run = [&](auto k4tag) {
    constexpr bool K4 = decltype(k4tag)::value;
    auto load_tile = [&](int k0) {
      const int k = k0 + akq;
      if (K4) {
        const unsigned tail = k < a.K ? 0u : OOB;           // OR-ed in: stays branch-free (a select here became an exec-masked branch)
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
      if (VEC) {
#pragma unroll
        for (int i = 0; i < NBV; ++i) {
          const int kr = k0 + i * VROWS;                         // + vrow0 is folded into vbase
          rv[i] = bload4(xr, (kr + vrow0 < a.K) ? vbase : OOB, kr * HW_ * 4);
        }
      } else if (V3) {
        rcode = 0;
#pragma unroll
        for (int i = 0; i < NBV; ++i) {
          const int kr = k0 + vrow0 + i * VROWS;
          const int c = kr / 9, rs = kr - c * 9;
          const unsigned e = taps[rs * VG + vg];
          rcode |= (e & 3u) << (2 * i);
          rv[i] = bload4(xr, (kr < a.K) ? (e & ~3u) + (unsigned)(c * HW_ * 4) : OOB, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int kr = k0 + krow0 + i * KSTEP;               // wave-uniform when BN >= 64
          const int c = kr / KK, rs = kr - c * KK;
          const unsigned off = (KS == 1) ? off1 : taps[rs * BN + nl];
          if (BN >= 64) rb[VEC ? 0 : i] = bload(xr, (kr < a.K) ? off : OOB, c * HW_ * 4);
          else rb[(VEC || V3) ? 0 : i] = bload(xr, (kr < a.K) ? off + (unsigned)(c * HW_ * 4) : OOB, 0);   // (OOB + channel offset stays >= 2^31)
        }
      }
    };
    auto store_tile = [&](int buf) {
      if (AALL || arow < BM) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) As[buf][(arow + AROWS * i) * LDA + akq + j] = ra[i][j];
      }
      if (VEC) {
#pragma unroll
        for (int i = 0; i < NBV; ++i) *reinterpret_cast<float4*>(&Bs[buf][(vrow0 + i * VROWS) * BN + vg * 4]) = rv[i];
      } else if (V3) {
#pragma unroll
        for (int i = 0; i < NBV; ++i) {
          const unsigned code = (rcode >> (2 * i)) & 3u;
          float4 v = rv[i];
          if (code == 1u) v = make_float4(0.f, v.x, v.y, v.z);
          else if (code == 2u) v = make_float4(v.y, v.z, v.w, 0.f);
          *reinterpret_cast<float4*>(&Bs[buf][(vrow0 + i * VROWS) * BN + vg * 4]) = v;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NB; ++i) Bs[buf][(krow0 + i * KSTEP) * BN + nl] = rb[(VEC || V3) ? 0 : i];
      }
    };
    auto mma_tile = [&](int buf) {
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        float av[TM], bv[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) av[i] = As[buf][(wm * TM * 32 + i * 32 + (lane & 31)) * LDA + kk * 2 + (lane >> 5)];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[j] = Bs[buf][(kk * 2 + (lane >> 5)) * BN + wn * TN * 32 + j * 32 + (lane & 31)];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
    };
    const int KTall = (a.K + BK - 1) / BK;
    const int kt0 = __builtin_amdgcn_readfirstlane((int)((int64_t)ksplit * KTall / nsplit)), KT = __builtin_amdgcn_readfirstlane((int)((int64_t)(ksplit + 1) * KTall / nsplit));
    if (a.ilv && KT - kt0 >= 2) {
      // Interleaved schedule (round 5).  In the loop below a wave issues its loads, runs its MFMAs, THEN stores to LDS and waits at the barrier:
      // only the other resident waves can hide the last two phases, and with 2-4 of them per SIMD that leaves the matrix pipe ~50 % busy.
      // Here the staging work of a wave is issued in the shadow of its own MFMAs (a v_mfma_f32_32x32x2_f32 holds the pipe for 64 cycles, a k
      // step has TM * TN of them): the operand pieces of a slice (A float4 groups, B float4 groups / gathered elements) are spread over the k
      // steps; in its step a piece is stored from registers to the LDS buffer of slice kt + 1 and immediately RE-LOADED for slice kt + 2 into
      // the same registers, so every load has a whole iteration to land and no register is added.  MFMA operands are read one k step ahead.
      constexpr int G = BK / 2, NBP = (VEC || V3) ? NBV : NB, NP = NA + NBP;
      unsigned rc[V3 ? NBV : 1];
      auto load_a_piece = [&](int i, int k0) {
        const int k = k0 + akq;
        if (K4) {
          const float4 v = bload4(wr, abase[i] | (k < a.K ? 0u : OOB), k0 * 4);
          ra[i][0] = v.x; ra[i][1] = v.y; ra[i][2] = v.z; ra[i][3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) ra[i][j] = bload(wr, (k + j < a.K) ? abase[i] + 4u * j : OOB, k0 * 4);
        }
      };
      auto store_a_piece = [&](int i, int buf) {
        if (AALL || arow < BM) {
#pragma unroll
          for (int j = 0; j < 4; ++j) As[buf][(arow + AROWS * i) * LDA + akq + j] = ra[i][j];
        }
      };
      auto load_b_piece = [&](int i, int k0) {
        if (VEC) {
          const int kr = k0 + i * VROWS;
          rv[i] = bload4(xr, (kr + vrow0 < a.K) ? vbase : OOB, kr * HW_ * 4);
        } else if (V3) {
          const int kr = k0 + vrow0 + i * VROWS;
          const int c = kr / 9, rs = kr - c * 9;
          const unsigned e = taps[rs * VG + vg];
          rc[V3 ? i : 0] = e & 3u;
          rv[i] = bload4(xr, (kr < a.K) ? (e & ~3u) + (unsigned)(c * HW_ * 4) : OOB, 0);
        } else {
          const int kr = k0 + krow0 + i * KSTEP;
          const int c = kr / KK, rs = kr - c * KK;
          const unsigned off = (KS == 1) ? off1 : taps[rs * BN + nl];
          if (BN >= 64) rb[i] = bload(xr, (kr < a.K) ? off : OOB, c * HW_ * 4);
          else rb[i] = bload(xr, (kr < a.K) ? off + (unsigned)(c * HW_ * 4) : OOB, 0);
        }
      };
      auto store_b_piece = [&](int i, int buf) {
        if (VEC) {
          *reinterpret_cast<float4*>(&Bs[buf][(vrow0 + i * VROWS) * BN + vg * 4]) = rv[i];
        } else if (V3) {
          const unsigned code = rc[V3 ? i : 0];
          float4 v = rv[i];
          if (code == 1u) v = make_float4(0.f, v.x, v.y, v.z);
          else if (code == 2u) v = make_float4(v.y, v.z, v.w, 0.f);
          *reinterpret_cast<float4*>(&Bs[buf][(vrow0 + i * VROWS) * BN + vg * 4]) = v;
        } else {
          Bs[buf][(krow0 + i * KSTEP) * BN + nl] = rb[i];
        }
      };
      // slice kt0 -> LDS, slice kt0 + 1 -> registers
#pragma unroll
      for (int i = 0; i < NA; ++i) load_a_piece(i, kt0 * BK);
#pragma unroll
      for (int i = 0; i < NBP; ++i) load_b_piece(i, kt0 * BK);
#pragma unroll
      for (int i = 0; i < NA; ++i) store_a_piece(i, kt0 & 1);
#pragma unroll
      for (int i = 0; i < NBP; ++i) store_b_piece(i, kt0 & 1);
#pragma unroll
      for (int i = 0; i < NA; ++i) load_a_piece(i, (kt0 + 1) * BK);
#pragma unroll
      for (int i = 0; i < NBP; ++i) load_b_piece(i, (kt0 + 1) * BK);
      __syncthreads();
      for (int kt = kt0; kt + 1 < KT; ++kt) {
        const int buf = kt & 1, nbuf = buf ^ 1, k2 = (kt + 2) * BK;
        float av[2][TM], bv[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) av[0][i] = As[buf][(wm * TM * 32 + i * 32 + (lane & 31)) * LDA + (lane >> 5)];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[0][j] = Bs[buf][(lane >> 5) * BN + wn * TN * 32 + j * 32 + (lane & 31)];
#pragma unroll
        for (int kk = 0; kk < G; ++kk) {
          if (kk + 1 < G) {
#pragma unroll
            for (int i = 0; i < TM; ++i) av[(kk + 1) & 1][i] = As[buf][(wm * TM * 32 + i * 32 + (lane & 31)) * LDA + (kk + 1) * 2 + (lane >> 5)];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[(kk + 1) & 1][j] = Bs[buf][((kk + 1) * 2 + (lane >> 5)) * BN + wn * TN * 32 + j * 32 + (lane & 31)];
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1][i], bv[kk & 1][j], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int p = 0; p < NP; ++p)
            if ((p * G) / NP == kk) {
              if (p < NA) { store_a_piece(p < NA ? p : 0, nbuf); load_a_piece(p < NA ? p : 0, k2); }
              else { store_b_piece(p >= NA ? p - NA : 0, nbuf); load_b_piece(p >= NA ? p - NA : 0, k2); }
            }
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
      }
      mma_tile((KT - 1) & 1);
      return;
    }
    load_tile(kt0 * BK);
    store_tile(kt0 & 1);
    __syncthreads();
    for (int kt = kt0; kt + 1 < KT; ++kt) {
      load_tile((kt + 1) * BK);
      __builtin_amdgcn_sched_barrier(0);          // keep the gather issued BEFORE the MFMA loop (hipcc otherwise sinks it to the stores)
      mma_tile(kt & 1);
      __builtin_amdgcn_sched_barrier(0);
      store_tile((kt + 1) & 1);
      __syncthreads();
    }
    mma_tile((KT - 1) & 1);
  };
  if (k4) run(std::true_type{}); else run(std::false_type{});

  // epilogue: C/D layout col = lane&31 (pixel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (channel).
  // Uniform decisions (split partial / bias / addend / activation / interior tile) are hoisted out of the element loops.
  const bool partial = nsplit > 1;
  float* __restrict__ outp = partial ? a.ws + (size_t)ksplit * a.B * a.M * HoWo_ : y_;
  const bool has_bias = !partial && a.bias != nullptr, has_add = !partial && add_ != nullptr;
  const int epi = partial ? PRN_EPI_NONE : a.epi;
  const bool interior = (m0 + BM <= a.M) && (n0 + BN <= N_);
  if (a.wide_store && a.ystride == 1 && (HoWo_ & 3) == 0) {
    // Wide stores: an accumulator block holds, per lane, 16 channels of ONE pixel -- stored as is, that is 16 dword stores per
    // block, each wave instruction touching two 128-byte row segments.  Transposed through LDS (the operand buffers are free
    // now) a lane owns 4 consecutive pixels of one channel: 4 dwordx4 stores per block, 512 contiguous bytes per 8 lanes,
    // and bias / addend are applied on float4s.  (n0 and HoWo are multiples of 4: a pixel quad never straddles two images.)
    __syncthreads();                                       // every wave is done reading As / Bs
    float* cw = smem + wave * (32 * CPITCH);
    const int crow = lane >> 3, ccol = (lane & 7) * 4;      // this lane's quad: rows crow + 8 * q, pixels ccol .. ccol + 3
    // K-split sum folded into this kernel (a.cnt != null): partial tiles are stored with agent scope; the workgroup that
    // arrives LAST at the tile's counter sums the nsplit partials in split order (the order of reduce_epilogue_kernel, so
    // the result does not depend on which workgroup that is) and applies bias / addend / activation.
    const bool fuse = partial && a.cnt != nullptr;
    const unsigned total4 = (unsigned)(a.B * a.M * HoWo_) * 4u;              // bytes of one partial slice
    const __amdgpu_buffer_rsrc_t wsr = make_rsrc(a.ws, fuse ? (int)(total4 * (unsigned)nsplit) : 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nq = n0 + wn * TN * 32 + j * 32 + ccol;
      const bool nok = interior || nq < N_;
      const int bb = nq / HoWo_, p = nq - bb * HoWo_;
      const size_t base = (size_t)bb * a.M * HoWo_ + p;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) cw[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CPITCH + (lane & 31)] = acc[i][j][r];
        __builtin_amdgcn_wave_barrier();
        const int mb = m0 + wm * TM * 32 + i * 32 + crow;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = mb + 8 * q;
          float4 v = *reinterpret_cast<const float4*>(&cw[(crow + 8 * q) * CPITCH + ccol]);
          if (!nok || (!interior && m >= a.M)) continue;
          const size_t idx = base + (size_t)m * HoWo_;
          if (has_bias) { const float bm = a.bias[m]; v.x += bm; v.y += bm; v.z += bm; v.w += bm; }
          if (has_add) { const float4 t = *reinterpret_cast<const float4*>(add_ + idx); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
          if (epi == PRN_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          else if (epi == PRN_EPI_SIGMOID) { v.x = 1.f / (1.f + __expf(-v.x)); v.y = 1.f / (1.f + __expf(-v.y)); v.z = 1.f / (1.f + __expf(-v.z)); v.w = 1.f / (1.f + __expf(-v.w)); }
          if (fuse) bstore4_agent(wsr, (unsigned)ksplit * total4 + (unsigned)idx * 4u, v);
          else *reinterpret_cast<float4*>(outp + idx) = v;
        }
        __builtin_amdgcn_wave_barrier();                   // the pad is rewritten by the next block
      }
    }
    if (!fuse) return;
    __shared__ int last_arrival;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's partial stores have reached memory ...
    __syncthreads();                                       // ... and so have the other waves'
    if (tid == 0) {
      const unsigned prev = __hip_atomic_fetch_add(a.cnt + id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last_arrival = prev == (unsigned)nsplit - 1u;
      if (last_arrival) __hip_atomic_store(a.cnt + id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch on this stream
    }
    __syncthreads();
    if (!last_arrival) return;
    const bool f_bias = a.bias != nullptr, f_add = add_ != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nq = n0 + wn * TN * 32 + j * 32 + ccol;
      const bool nok = interior || nq < N_;
      const int bb = nq / HoWo_, p = nq - bb * HoWo_;
      const size_t base = (size_t)bb * a.M * HoWo_ + p;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mb = m0 + wm * TM * 32 + i * 32 + crow;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = mb + 8 * q;
          if (!nok || (!interior && m >= a.M)) continue;
          const size_t idx = base + (size_t)m * HoWo_;
          float4 v = bload4_agent(wsr, (unsigned)idx * 4u);
          for (int sp = 1; sp < nsplit; ++sp) {
            const float4 t = bload4_agent(wsr, (unsigned)sp * total4 + (unsigned)idx * 4u);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
          }
          if (f_bias) { const float bm = a.bias[m]; v.x += bm; v.y += bm; v.z += bm; v.w += bm; }
          if (f_add) { const float4 t = *reinterpret_cast<const float4*>(add_ + idx); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
          if (a.epi == PRN_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          else if (a.epi == PRN_EPI_SIGMOID) { v.x = 1.f / (1.f + __expf(-v.x)); v.y = 1.f / (1.f + __expf(-v.y)); v.z = 1.f / (1.f + __expf(-v.z)); v.w = 1.f / (1.f + __expf(-v.w)); }
          *reinterpret_cast<float4*>(y_ + idx) = v;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nn = n0 + wn * TN * 32 + j * 32 + (lane & 31);
    if (!interior && nn >= N_) continue;
    const int bb = nn / HoWo_, p = nn - bb * HoWo_;
    size_t base = (size_t)bb * a.M * HoWo_ + p;
    size_t mstride = HoWo_;
    if (a.ystride != 1) {                               // strided / phase-interleaved output plane (never with a K split)
      const int qh = p / Wo_, qw = p - qh * Wo_;
      mstride = a.yHW;
      base = (size_t)bb * a.M * a.yHW + (size_t)(qh * a.ystride + py) * a.yW + qw * a.ystride + px;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mbase = m0 + wm * TM * 32 + i * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2);
        if (!interior && m >= a.M) continue;
        const size_t idx = base + (size_t)m * mstride;
        float v = acc[i][j][r];
        if (has_bias) v += a.bias[m];
        if (has_add) v += add_[idx];
        if (epi == PRN_EPI_RELU) v = fmaxf(v, 0.f);
        else if (epi == PRN_EPI_SIGMOID) v = 1.f / (1.f + __expf(-v));
        outp[idx] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- wgrad
struct WgArgs {
  const float* x; const float* dy; float* out;
  int B, C, H, W, M, stride, pad, Ho, Wo;
  int K, N, HoWo, HW, tilesM, tilesJ, splits, chunks;
  int xbytes, dybytes;       // dybytes: one phase of dy
  int xz;                    // element stride of x per blockIdx.z (prn_gemm_batched_nt; 0 for a convolution)
  int nzg;                   // grid depth: phases / batched products / layers of a group (the third grid dimension, folded into the 1-D grid)
  int ngroup;                // > 0: blockIdx.z = layer of a group of same-shape layers (prn_conv2d_wgrad_grouped): x / dy from the tables below
  int ilv;                   // interleaved reduction loop (conv_igemm_kernel has the description)
  const float* gx[PRN_WGRAD_GROUP_MAX];
  const float* gdy[PRN_WGRAD_GROUP_MAX];
  Seg seg;
};

// entry `i` of one of the pointer tables of the by-value kernel argument, read with scalar loads (see seg_read)
template <class Args>
__device__ __forceinline__ const float* kernarg_ptr(size_t member_offset, int i) {
  typedef __attribute__((address_space(4))) const uint64_t* kptr;
  kptr base = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
  return reinterpret_cast<const float*>(base[member_offset / 8 + i]);
}

template <int KS, int MODE, int TM, int TJ, int WM = 2, bool RAG = false>      // RAG: ragged-batch instance (keeps the dense ones at 152 VGPRs)
__global__ __launch_bounds__(256, (RAG && TM * TJ == 4) ? 3 : 1) void conv_wgrad_kernel(WgArgs a) {
  constexpr int WJ = 4 / WM;                     // WM = 1: 32 x 128 tile for <= 32 output channels (see conv_igemm_kernel)
  constexpr int BM = 32 * WM * TM, BJ = 32 * WJ * TJ, LD = 17, KK = KS * KS;
  constexpr int NBJ = BJ / 16;
  // one LDS allocation: the operand double buffers during the reduction, then (epilogue) one 32 x 36 transpose pad per wave
  constexpr int AS_ = 2 * BM * LD, BS_ = 2 * BJ * LD, CS_ = 4 * 32 * 36;
  __shared__ __attribute__((aligned(16))) float smem_[(AS_ + BS_) > CS_ ? (AS_ + BS_) : CS_];
  float (*As)[BM * LD] = reinterpret_cast<float (*)[BM * LD]>(smem_);
  float (*Bs)[BJ * LD] = reinterpret_cast<float (*)[BJ * LD]>(smem_ + AS_);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WJ, wj = wave % WJ;
  // 1-D grid renumbered XCD-aware: the output tiles of one (pixel split, z) stream the same dy / x pixels and sit on one XCD's L2
  // (see prn_wgrad16.hip; as a (tile, split, z) grid the tiles of a split were dealt out round-robin over the eight L2s)
  const int ntile_ = a.tilesM * a.tilesJ;
  const int lid_ = prn_xcd_remap(blockIdx.x, ntile_ * a.splits * a.nzg);
  const int tile_ = lid_ % ntile_, rest_ = lid_ / ntile_;
  const int m0 = (tile_ % a.tilesM) * BM, j0 = (tile_ / a.tilesM) * BJ;
  const int split = rest_ % a.splits;
  const int bz = rest_ / a.splits;               // phase / batch index / layer of the group (was bz)
  const int cbeg = (int)((int64_t)split * a.chunks / a.splits), cend = (int)((int64_t)(split + 1) * a.chunks / a.splits);

  // PRN_IN_UP2_PHASE: bz = phase; dy is phase-major [4][B][M][H][W] (prn_space_to_depth2), out is [4][M][4C]
  const int py = (MODE == PRN_IN_UP2_PHASE) ? (int)(bz >> 1) : 0, px = (MODE == PRN_IN_UP2_PHASE) ? (int)(bz & 1) : 0;
  const int pad_y = (MODE == PRN_IN_UP2_PHASE) ? 1 - py : a.pad, pad_x = (MODE == PRN_IN_UP2_PHASE) ? 1 - px : a.pad;
  const float* dyz = a.ngroup > 0 ? kernarg_ptr<WgArgs>(offsetof(WgArgs, gdy), bz) : a.dy + (size_t)bz * (a.dybytes / 4);
  const float* xz_ = a.ngroup > 0 ? kernarg_ptr<WgArgs>(offsetof(WgArgs, gx), bz) : a.x + (size_t)bz * a.xz;
  int H_ = a.H, W_ = a.W, HW_ = a.HW, Ho_ = a.Ho, Wo_ = a.Wo, HoWo_ = a.HoWo, N_ = a.N;     // current segment's geometry (ragged) / the tensor's
  __amdgpu_buffer_rsrc_t xr = make_rsrc(xz_, a.xbytes), dyr = make_rsrc(dyz, a.dybytes);
  const int arow = tid >> 2, anq = (tid & 3) * 4;
  const bool n4 = ((HoWo_ & 3) == 0 || a.seg.nseg > 0) && ((reinterpret_cast<uintptr_t>(dyz) & 15) == 0);   // (ragged: host checked every segment)
  const int nl = tid & 15, jrow = tid >> 4;
  int jcoff[NBJ], jr[NBJ], js[NBJ];
  bool jok[NBJ];
#pragma unroll
  for (int i = 0; i < NBJ; ++i) {
    const int j = j0 + jrow + 16 * i;
    const int c = j / KK, rs = j - c * KK;
    jok[i] = j < a.K;
    jcoff[i] = jok[i] ? c * HW_ : 0;
    jr[i] = rs / KS;
    js[i] = rs - jr[i] * KS;
  }
  bool mok[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) mok[i] = (m0 + arow + 64 * i < a.M) && (arow + 64 * i < BM);

  float ra[TM][4], rb[NBJ];
  f32x16 acc[TM][TJ];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // pixel cursors of this thread (dY column group and im2col pixel), advanced by 16 pixels per chunk -- no divisions in the loop
  int a_b, a_p, g_b, g_oh, g_ow;
  int nbase = 0, nend = N_;                              // pixel range of the current segment on the GEMM's pixel axis
  auto enter = [&](int ch) {                              // (re)initialise geometry and cursors at chunk `ch`
    if (RAG) {
      const int sg = seg_find(a.seg, ch * 16);
      H_ = Ho_ = SEG_READ(WgArgs, H, sg);
      W_ = Wo_ = SEG_READ(WgArgs, W, sg);
      HW_ = HoWo_ = H_ * W_;
      nbase = SEG_READ(WgArgs, n0, sg);
      N_ = a.B * HoWo_;
      nend = nbase + N_;
      xr = make_rsrc(a.x + SEG_READ(WgArgs, x0, sg), a.B * a.C * HW_ * 4);
      dyr = make_rsrc(dyz + SEG_READ(WgArgs, y0, sg), a.B * a.M * HoWo_ * 4);
#pragma unroll
      for (int i = 0; i < NBJ; ++i) jcoff[i] = jok[i] ? ((j0 + jrow + 16 * i) / KK) * HW_ : 0;
    }
    const int n = ch * 16 - nbase + anq;
    a_b = n / HoWo_; a_p = n - a_b * HoWo_;
    const int g = ch * 16 - nbase + nl;
    g_b = g / HoWo_;
    const int p = g - g_b * HoWo_;
    g_oh = p / Wo_; g_ow = p - g_oh * Wo_;
  };
  enter(cbeg);
  // every fetch is a buffer load whose offset is OOB (-> 0.0) for elements that do not exist; nothing depends on the
  // loaded values until store_chunk, so the loads overlap the MFMA loop of the previous chunk.  N4 is a compile-time
  // tag and the last chunk is peeled (branch-free loop body, see conv_igemm_kernel).
  auto run = [&](auto n4tag) {
    constexpr bool N4 = decltype(n4tag)::value;
    auto load_chunk = [&](int ch) {
      if (RAG && ch * 16 >= nend) enter(ch);              // ragged batch: next segment (uniform, a handful of times per workgroup)
      {  // dY rows: 4 consecutive pixels of one output channel
        const int n = ch * 16 - nbase + anq;
        if (N4) {
          const bool ok = n < N_;
          const unsigned base = (unsigned)((a_b * a.M + m0 + arow) * HoWo_ + a_p) * 4u;
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const float4 v = bload4(dyr, (ok && mok[i]) ? base : OOB, i * 64 * HoWo_ * 4);
            ra[i][0] = v.x; ra[i][1] = v.y; ra[i][2] = v.z; ra[i][3] = v.w;
          }
        } else {
          int qb = a_b, qp = a_p;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bool ok = n + q < N_;
            const unsigned base = (unsigned)((qb * a.M + m0 + arow) * HoWo_ + qp) * 4u;
#pragma unroll
            for (int i = 0; i < TM; ++i) ra[i][q] = bload(dyr, (ok && mok[i]) ? base : OOB, i * 64 * HoWo_ * 4);
            if (++qp >= HoWo_) { qp = 0; ++qb; }
          }
        }
        a_p += 16;
        while (a_p >= HoWo_) { a_p -= HoWo_; ++a_b; }
      }
      {  // im2col rows
        const bool ok = ch * 16 - nbase + nl < N_;
        const int ih0 = g_oh * a.stride - pad_y, iw0 = g_ow * a.stride - pad_x;
        const int pix0 = g_b * a.C * HW_;
#pragma unroll
        for (int i = 0; i < NBJ; ++i) {
          const int off = tap_offset<MODE>(ih0 + jr[i], iw0 + js[i], ok && jok[i], H_, W_);
          rb[i] = bload(xr, off >= 0 ? (unsigned)(pix0 + jcoff[i] + off) * 4u : OOB, 0);
        }
        g_ow += 16;
        while (g_ow >= Wo_) { g_ow -= Wo_; ++g_oh; }
        while (g_oh >= Ho_) { g_oh -= Ho_; ++g_b; }
      }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        if (BM >= 64 || arow < BM) {
#pragma unroll
          for (int q = 0; q < 4; ++q) As[buf][(arow + 64 * i) * LD + anq + q] = ra[i][q];
        }
#pragma unroll
      for (int i = 0; i < NBJ; ++i) Bs[buf][(jrow + 16 * i) * LD + nl] = rb[i];
    };
    auto mma_chunk = [&](int buf) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        float av[TM], bv[TJ];
#pragma unroll
        for (int i = 0; i < TM; ++i) av[i] = As[buf][(wm * TM * 32 + i * 32 + (lane & 31)) * LD + kk * 2 + (lane >> 5)];
#pragma unroll
        for (int j = 0; j < TJ; ++j) bv[j] = Bs[buf][(wj * TJ * 32 + j * 32 + (lane & 31)) * LD + kk * 2 + (lane >> 5)];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
    };
    if (cbeg >= cend) return;
    if (N4 && !RAG && a.ilv && cend - cbeg >= 2) {
      // Interleaved schedule (see conv_igemm_kernel): a piece of the next chunk's operands (a dY float4 group, a gathered im2col element) is
      // stored to LDS and re-loaded for the chunk after it in the shadow of each k step's MFMAs; operands are read one k step ahead.
      constexpr int NP = TM + NBJ;
      load_chunk(cbeg);
      store_chunk(0);
      load_chunk(cbeg + 1);
      __syncthreads();
      for (int ch = cbeg; ch + 1 < cend; ++ch) {
        const int buf = (ch - cbeg) & 1, nbuf = buf ^ 1;
        // cursors stand at chunk ch + 2
        const bool aok = (ch + 2) * 16 + anq < N_;
        const unsigned abase2 = (unsigned)((a_b * a.M + m0 + arow) * HoWo_ + a_p) * 4u;
        const bool bok = (ch + 2) * 16 + nl < N_;
        const int ih0 = g_oh * a.stride - pad_y, iw0 = g_ow * a.stride - pad_x;
        const int pix0 = g_b * a.C * HW_;
        float av[2][TM], bv[2][TJ];
#pragma unroll
        for (int i = 0; i < TM; ++i) av[0][i] = As[buf][(wm * TM * 32 + i * 32 + (lane & 31)) * LD + (lane >> 5)];
#pragma unroll
        for (int j = 0; j < TJ; ++j) bv[0][j] = Bs[buf][(wj * TJ * 32 + j * 32 + (lane & 31)) * LD + (lane >> 5)];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          if (kk + 1 < 8) {
#pragma unroll
            for (int i = 0; i < TM; ++i) av[(kk + 1) & 1][i] = As[buf][(wm * TM * 32 + i * 32 + (lane & 31)) * LD + (kk + 1) * 2 + (lane >> 5)];
#pragma unroll
            for (int j = 0; j < TJ; ++j) bv[(kk + 1) & 1][j] = Bs[buf][(wj * TJ * 32 + j * 32 + (lane & 31)) * LD + (kk + 1) * 2 + (lane >> 5)];
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1][i], bv[kk & 1][j], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int p = 0; p < NP; ++p)
            if ((p * 8) / NP == kk) {
              if (p < TM) {
                const int i = p < TM ? p : 0;
                if (BM >= 64 || arow < BM) {
#pragma unroll
                  for (int q = 0; q < 4; ++q) As[nbuf][(arow + 64 * i) * LD + anq + q] = ra[i][q];
                }
                const float4 v = bload4(dyr, (aok && mok[i]) ? abase2 : OOB, i * 64 * HoWo_ * 4);
                ra[i][0] = v.x; ra[i][1] = v.y; ra[i][2] = v.z; ra[i][3] = v.w;
              } else {
                const int i = p >= TM ? p - TM : 0;
                Bs[nbuf][(jrow + 16 * i) * LD + nl] = rb[i];
                const int off = tap_offset<MODE>(ih0 + jr[i], iw0 + js[i], bok && jok[i], H_, W_);
                rb[i] = bload(xr, off >= 0 ? (unsigned)(pix0 + jcoff[i] + off) * 4u : OOB, 0);
              }
            }
          __builtin_amdgcn_sched_barrier(0);
        }
        a_p += 16;
        while (a_p >= HoWo_) { a_p -= HoWo_; ++a_b; }
        g_ow += 16;
        while (g_ow >= Wo_) { g_ow -= Wo_; ++g_oh; }
        while (g_oh >= Ho_) { g_oh -= Ho_; ++g_b; }
        __syncthreads();
      }
      mma_chunk((cend - 1 - cbeg) & 1);
      return;
    }
    load_chunk(cbeg);
    store_chunk(0);
    __syncthreads();
    for (int ch = cbeg; ch + 1 < cend; ++ch) {
      load_chunk(ch + 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_chunk((ch - cbeg) & 1);
      __builtin_amdgcn_sched_barrier(0);
      store_chunk((ch + 1 - cbeg) & 1);
      __syncthreads();
    }
    mma_chunk((cend - 1 - cbeg) & 1);
  };
  if (n4) run(std::true_type{}); else run(std::false_type{});

  float* out = a.out + ((size_t)split * a.nzg + bz) * a.M * a.K;
  if ((a.K & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0) && (((size_t)a.M * a.K) & 3) == 0) {
    // epilogue through an LDS transpose (the operand buffers are dead): dwordx4 stores of four consecutive dW columns per lane instead of
    // 16 dword stores per accumulator block (as in conv_igemm_kernel / split16_gemm_kernel)
    __syncthreads();
    float* cw = smem_ + wave * (32 * 36);
    const int crow = lane >> 3, ccol = (lane & 7) * 4;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int jj4 = j0 + wj * TJ * 32 + j * 32 + ccol;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) cw[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = acc[i][j][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = m0 + wm * TM * 32 + i * 32 + crow + 8 * q;
          const float4 v = *reinterpret_cast<const float4*>(&cw[(crow + 8 * q) * 36 + ccol]);
          if (m < a.M && jj4 < a.K) *reinterpret_cast<float4*>(out + (size_t)m * a.K + jj4) = v;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int jj = j0 + wj * TJ * 32 + j * 32 + (lane & 31);
    if (jj >= a.K) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < a.M) out[(size_t)m * a.K + jj] = acc[i][j][r];
      }
  }
}

// ---------------------------------------------------------------------------------------------- direct small kernels
// 3x3 stride-1 layers with one or two output channels over a large map and few input channels (the depth head: 64 -> 1 at
// 240x320), and their input gradients (1 -> 64), are HBM-bound streams, not GEMMs: on the 64-row MFMA tile they ran at 2-4
// TFLOP/s (253 / 183 / 416 us for forward / input gradient / weight gradient).  One thread per pixel instead.
template <int MODE, int MM>
__global__ __launch_bounds__(256) void conv3x3_small_m_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                              const float* __restrict__ addend, float* __restrict__ y, int B, int C, int H, int W,
                                                              int pad, int epi) {
  const int HW = H * W;
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= (int64_t)B * HW) return;
  const int b = (int)(n / HW), p = (int)(n - (int64_t)b * HW), oh = p / W, ow = p - oh * W;
  int off[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) off[t] = tap_offset<MODE>(oh - pad + t / 3, ow - pad + t % 3, true, H, W);
  float acc[MM];
#pragma unroll
  for (int m = 0; m < MM; ++m) acc[m] = 0.f;
  const float* xb = x + (size_t)b * C * HW;
  for (int c = 0; c < C; ++c) {                            // same summation order as the GEMM's K axis (c major, tap minor)
    float v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) v[t] = off[t] >= 0 ? xb[(size_t)c * HW + off[t]] : 0.f;
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[m] += w[((size_t)m * C + c) * 9 + t] * v[t];
  }
#pragma unroll
  for (int m = 0; m < MM; ++m) {
    const size_t idx = ((size_t)b * MM + m) * HW + p;
    float r = acc[m];
    if (bias) r += bias[m];
    if (addend) r += addend[idx];
    if (epi == PRN_EPI_RELU) r = fmaxf(r, 0.f);
    else if (epi == PRN_EPI_SIGMOID) r = 1.f / (1.f + __expf(-r));
    y[idx] = r;
  }
}

// one input channel -> M output channels, zero padding: the 9 taps of a pixel are read once and every output row is a
// 9-term dot product; the launch is bound by writing y
__global__ __launch_bounds__(256) void conv3x3_one_c_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            const float* __restrict__ addend, float* __restrict__ y, int B, int H, int W, int M,
                                                            int pad, int Ho, int Wo, int epi) {
  const int HoWo = Ho * Wo;
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= (int64_t)B * HoWo) return;
  const int b = (int)(n / HoWo), p = (int)(n - (int64_t)b * HoWo), oh = p / Wo, ow = p - oh * Wo;
  const float* xb = x + (size_t)b * H * W;
  float v[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int o = tap_offset<PRN_IN_ZERO>(oh - pad + t / 3, ow - pad + t % 3, true, H, W);
    v[t] = o >= 0 ? xb[o] : 0.f;
  }
  for (int m = 0; m < M; ++m) {
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) r += w[(size_t)m * 9 + t] * v[t];
    const size_t idx = ((size_t)b * M + m) * HoWo + p;
    if (bias) r += bias[m];
    if (addend) r += addend[idx];
    if (epi == PRN_EPI_RELU) r = fmaxf(r, 0.f);
    else if (epi == PRN_EPI_SIGMOID) r = 1.f / (1.f + __expf(-r));
    y[idx] = r;
  }
}

// weight gradient of the one-output-channel layer: block (c, split) reduces its pixel range for the 9 taps of channel c
template <int MODE>
__global__ __launch_bounds__(256) void conv3x3_small_m_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out,
                                                                    int B, int C, int H, int W, int M, int pad) {
  const int c = blockIdx.x % C, m = blockIdx.x / C, sp = blockIdx.y, S = gridDim.y;
  const int HW = H * W;
  const int64_t N = (int64_t)B * HW, beg = N * sp / S, end = N * (sp + 1) / S;
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  for (int64_t n = beg + threadIdx.x; n < end; n += 256) {
    const int b = (int)(n / HW), p = (int)(n - (int64_t)b * HW), oh = p / W, ow = p - oh * W;
    const float g = dy[((size_t)b * M + m) * HW + p];
    const float* xc = x + ((size_t)b * C + c) * HW;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int o = tap_offset<MODE>(oh - pad + t / 3, ow - pad + t % 3, true, H, W);
      acc[t] += g * (o >= 0 ? xc[o] : 0.f);
    }
  }
  __shared__ float sm[4][9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float v = acc[t];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][t] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const int t = threadIdx.x;
    out[((size_t)sp * M + m) * C * 9 + (size_t)c * 9 + t] = (sm[0][t] + sm[1][t]) + (sm[2][t] + sm[3][t]);
  }
}

// y = epi( sum_s ws[s] + bias[m] + addend ) for split-K launches (fixed summation order).
// (Folding this into the GEMM kernel -- last workgroup to arrive at a per-tile counter sums the partials: with a device-scope
// __threadfence() as the release / acquire pair it was 4x SLOWER, 57 -> 257 us on 1x1 1024->256, because the fence writes back and
// invalidates the XCD's whole L2; with agent-scope stores / loads instead of fences it works, see fused_reduce_ok.)
__global__ __launch_bounds__(256) void reduce_epilogue_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                                               const float* __restrict__ addend, float* __restrict__ y, int64_t total,
                                                               int M, int HoWo, int splits, int epi) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= total) return;
  if (i + 3 < total && (HoWo & 3) == 0) {
    float4 v = *reinterpret_cast<const float4*>(ws + i);
    for (int s = 1; s < splits; ++s) {
      const float4 t = *reinterpret_cast<const float4*>(ws + (size_t)s * total + i);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const int m = (int)((i / HoWo) % M);
    if (bias) { const float bm = bias[m]; v.x += bm; v.y += bm; v.z += bm; v.w += bm; }
    if (addend) { const float4 t = *reinterpret_cast<const float4*>(addend + i); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    if (epi == PRN_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    else if (epi == PRN_EPI_SIGMOID) { v.x = 1.f / (1.f + __expf(-v.x)); v.y = 1.f / (1.f + __expf(-v.y)); v.z = 1.f / (1.f + __expf(-v.z)); v.w = 1.f / (1.f + __expf(-v.w)); }
    *reinterpret_cast<float4*>(y + i) = v;
  } else {
    for (int64_t q = i; q < total && q < i + 4; ++q) {
      float v = ws[q];
      for (int s = 1; s < splits; ++s) v += ws[(size_t)s * total + q];
      if (bias) v += bias[(q / HoWo) % M];
      if (addend) v += addend[q];
      if (epi == PRN_EPI_RELU) v = fmaxf(v, 0.f);
      else if (epi == PRN_EPI_SIGMOID) v = 1.f / (1.f + __expf(-v));
      y[q] = v;
    }
  }
}

// The same for the pixel range [n_start, N) only (tail split of conv_igemm_kernel): one thread per (channel m, 4 pixels).
__global__ __launch_bounds__(256) void reduce_tail_kernel(const float* __restrict__ ws, const float* __restrict__ bias, const float* __restrict__ addend,
                                                          float* __restrict__ y, int64_t total, int M, int HoWo, int n_start, int ntail4, int splits,
                                                          int epi) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)M * ntail4) return;
  const int m = (int)(t / ntail4), q = (int)(t - (int64_t)m * ntail4);
  const int n = n_start + q * 4, b = n / HoWo, p = n - b * HoWo;
  const int64_t i = ((int64_t)b * M + m) * HoWo + p;
  float4 v = *reinterpret_cast<const float4*>(ws + i);
  for (int s = 1; s < splits; ++s) {
    const float4 u = *reinterpret_cast<const float4*>(ws + (size_t)s * total + i);
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  if (bias) { const float bm = bias[m]; v.x += bm; v.y += bm; v.z += bm; v.w += bm; }
  if (addend) { const float4 u = *reinterpret_cast<const float4*>(addend + i); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
  if (epi == PRN_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  else if (epi == PRN_EPI_SIGMOID) { v.x = 1.f / (1.f + __expf(-v.x)); v.y = 1.f / (1.f + __expf(-v.y)); v.z = 1.f / (1.f + __expf(-v.z)); v.w = 1.f / (1.f + __expf(-v.w)); }
  *reinterpret_cast<float4*>(y + i) = v;
}

// out[i] = sum_k ws[k][i] in a fixed order: four wave-sized groups each take every 4th split (four loads in flight per
// thread), then one LDS step adds the four partials.  n/64 workgroups instead of n/256 single-chain threads: the weight
// matrices are small (3e4 .. 2e6 elements), so the one-thread-per-element version left most CUs idle and latency-bound.
__global__ __launch_bounds__(256) void reduce_splits_kernel(const float* __restrict__ ws, float* __restrict__ out, int64_t n, int splits) {
  __shared__ float sm[4][64];
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + e;
  float acc = 0.f;
  if (i < n) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int k = g;
    for (; k + 12 < splits; k += 16) {
      a0 += ws[(size_t)k * n + i];
      a1 += ws[(size_t)(k + 4) * n + i];
      a2 += ws[(size_t)(k + 8) * n + i];
      a3 += ws[(size_t)(k + 12) * n + i];
    }
    for (; k < splits; k += 4) a0 += ws[(size_t)k * n + i];
    acc = (a0 + a1) + (a2 + a3);
  }
  sm[g][e] = acc;
  __syncthreads();
  if (g == 0 && i < n) out[i] = (sm[0][e] + sm[1][e]) + (sm[2][e] + sm[3][e]);
}

__global__ void flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int M, int C, int KH, int KW) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // index into wt [C][M][KH][KW]
  const int64_t total = (int64_t)M * C * KH * KW;
  if (i >= total) return;
  const int s = i % KW, r = (i / KW) % KH, m = (i / (KW * KH)) % M, c = i / ((int64_t)KW * KH * M);
  wt[i] = w[(((size_t)m * C + c) * KH + (KH - 1 - r)) * KW + (KW - 1 - s)];
}

// the same permutation for a whole list of weight tensors in one launch (one item per tensor): a training step flips ~180
// weights, each a 5 us launch otherwise.  A workgroup moves one 32 (m) x 32 (c) block of one tensor, all KH*KW taps, through
// LDS: rows of w are read along (c, tap) and rows of wt written along (m, tap), both contiguous -- the element-wise version
// read with stride C*KH*KW and fetched 6x the bytes it needed (PMC FETCH_SIZE: 1.0 GB for 172 MB of weights, 0.53 ms).
// `first` = running count of 32x32 blocks of the preceding items; kernels larger than 3x3 take the element-wise path.
// (LDS: a 32 x 32 block of a 1x1 weight, or ONE 16 x 16 quarter of a 3x3 block at a time -- 9.3 KB instead of the 37 KB of a whole 3x3 block, which capped every workgroup
//  of the launch at four per CU: 175 -> 161 us for 460 MB.  Round 5 also tried several blocks per workgroup with one bisection of the item table in the two split-image
//  kernels: 120 -> 122 / 108 us, i.e. neither the bisection nor the occupancy is what holds these three passes at 1.5-3 TB/s; not kept)
__global__ __launch_bounds__(256) void flip_transpose_batched_kernel(const prn_flip_item* __restrict__ items, int n, int64_t total) {
  __shared__ float tile[16 * (16 * 9 + 1)];
  const int64_t blk = blockIdx.x;
  if (blk >= total) return;
  int lo = 0, hi = n - 1;                                  // last item with first <= blk
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].first <= blk) lo = mid; else hi = mid - 1;
  }
  const prn_flip_item it = items[lo];
  const int KK = it.KH * it.KW;
  const int tilesC = (it.C + 31) / 32;
  const int t = (int)(blk - it.first), m0 = (t / tilesC) * 32, c0 = (t % tilesC) * 32;
  const int mt = min(32, it.M - m0), ct = min(32, it.C - c0);
  if (KK > 9) {                                            // 7x7 stem: tiny, element-wise
    for (int i = threadIdx.x; i < mt * ct * KK; i += 256) {
      const int k = i % KK, m = (i / KK) % mt, c = i / (KK * mt);
      it.dst[((size_t)(c0 + c) * it.M + m0 + m) * KK + k] = it.src[((size_t)(m0 + m) * it.C + c0 + c) * KK + (KK - 1 - k)];
    }
    return;
  }
  const int sub = KK == 1 ? 32 : 16;                         // (m, c) extent of one pass through LDS
  const int pitch = sub * KK + 1;
  for (int ms = 0; ms < mt; ms += sub)
    for (int cs = 0; cs < ct; cs += sub) {
      const int mq = min(sub, mt - ms), cq = min(sub, ct - cs);
      const int rowlen = cq * KK;
      __syncthreads();                                       // (the previous pass has been written out)
      for (int i = threadIdx.x; i < mq * rowlen; i += 256) { // read: row m of w, columns (c0+cs .. +cq) x taps, contiguous
        const int m = i / rowlen, j = i - m * rowlen;
        tile[m * pitch + j] = it.src[((size_t)(m0 + ms + m) * it.C + c0 + cs) * KK + j];
      }
      __syncthreads();
      const int orow = mq * KK;
      for (int i = threadIdx.x; i < cq * orow; i += 256) {   // write: row c of wt, columns (m0+ms .. +mq) x flipped taps, contiguous
        const int c = i / orow, j = i - c * orow, m = j / KK, k = j - m * KK;
        it.dst[((size_t)(c0 + cs + c) * it.M + m0 + ms) * KK + j] = tile[m * pitch + c * KK + (KK - 1 - k)];
      }
    }
}

// dx[b,c,h,w] = sum over the virtual padded positions that gather from (h,w)
__device__ __forceinline__ float pad_fold_one(const float* __restrict__ p, int h, int w, int Hv, int Wv, int Wp, int up2) {
  float acc = 0.f;
  const int u0 = up2 ? 2 * h : h, u1 = up2 ? 2 * h + 1 : h, v0 = up2 ? 2 * w : w, v1 = up2 ? 2 * w + 1 : w;
  for (int u = u0; u <= u1; ++u) {
    // padded rows that read virtual row u: u+1 always; 0 if u == 1; Hv+1 if u == Hv-2
    int rows[3], nr = 0;
    rows[nr++] = u + 1;
    if (u == 1) rows[nr++] = 0;
    if (u == Hv - 2) rows[nr++] = Hv + 1;
    for (int v = v0; v <= v1; ++v) {
      int cols[3], nc = 0;
      cols[nc++] = v + 1;
      if (v == 1) cols[nc++] = 0;
      if (v == Wv - 2) cols[nc++] = Wv + 1;
      for (int a = 0; a < nr; ++a)
        for (int c = 0; c < nc; ++c) acc += p[(int64_t)rows[a] * Wp + cols[c]];
    }
  }
  return acc;
}
__global__ __launch_bounds__(256) void pad_fold_kernel(const float* __restrict__ dp, float* __restrict__ dx, int BC, int H, int W, int up2, int pitch) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)BC * H * W) return;
  const int w = i % W, h = (i / W) % H;
  const int Hv = up2 ? 2 * H : H, Wv = up2 ? 2 * W : W;     // virtual (pre-pad) size
  const int Wp = pitch > 0 ? pitch : Wv + 2;
  dx[i] = pad_fold_one(dp + (i / ((int64_t)W * H)) * (int64_t)(Hv + 2) * Wp, h, w, Hv, Wv, Wp, up2);
}
// The same for W % 4 == 0, four consecutive pixels per thread.  A quad away from the border rows / columns has ONE padded position per virtual
// one: a single 16-byte load (or, under the x2 upsample, two per row of the 2 x 2 footprints: out = (((0 + a) + b) + c) + d in the loop's order)
// at a 4-byte aligned address, one aligned 16-byte store.  Quads next to the border run the loop per pixel.  (One pixel per thread moved 4 bytes
// per memory instruction and lane: 1.9-2.3 TB/s on the decoder's gradients.)
__global__ __launch_bounds__(256) void pad_fold4_kernel(const float* __restrict__ dp, float* __restrict__ dx, int BC, int H, int W, int up2, int pitch) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int W4 = W >> 2;
  if (i >= (int64_t)BC * H * W4) return;
  const int w = (int)(i % W4) * 4, h = (i / W4) % H;
  const int64_t bc = i / ((int64_t)H * W4);
  const int Hv = up2 ? 2 * H : H, Wv = up2 ? 2 * W : W;
  const int Wp = pitch > 0 ? pitch : Wv + 2;
  const float* p = dp + bc * (int64_t)(Hv + 2) * Wp;
  const bool colb = w == 0 || w == W - 4;                    // the quad holds virtual column 1 / Wv - 2
  float4 o;
  if (up2 && !(h == 0 || h == H - 1 || colb)) {
    const float* q = p + (int64_t)(2 * h + 1) * Wp + (2 * w + 1);
    const prn_f4u a0 = *reinterpret_cast<const prn_f4u*>(q), a1 = *reinterpret_cast<const prn_f4u*>(q + 4);
    const prn_f4u b0 = *reinterpret_cast<const prn_f4u*>(q + Wp), b1 = *reinterpret_cast<const prn_f4u*>(q + Wp + 4);
    o.x = (((0.f + a0.v[0]) + a0.v[1]) + b0.v[0]) + b0.v[1];
    o.y = (((0.f + a0.v[2]) + a0.v[3]) + b0.v[2]) + b0.v[3];
    o.z = (((0.f + a1.v[0]) + a1.v[1]) + b1.v[0]) + b1.v[1];
    o.w = (((0.f + a1.v[2]) + a1.v[3]) + b1.v[2]) + b1.v[3];
  } else if (!up2 && !(h == 1 || h == Hv - 2 || colb)) {
    const prn_f4u a = *reinterpret_cast<const prn_f4u*>(p + (int64_t)(h + 1) * Wp + (w + 1));
    o = make_float4(0.f + a.v[0], 0.f + a.v[1], 0.f + a.v[2], 0.f + a.v[3]);
  } else {
    o.x = pad_fold_one(p, h, w, Hv, Wv, Wp, up2); o.y = pad_fold_one(p, h, w + 1, Hv, Wv, Wp, up2);
    o.z = pad_fold_one(p, h, w + 2, Hv, Wv, Wp, up2); o.w = pad_fold_one(p, h, w + 3, Hv, Wv, Wp, up2);
  }
  *reinterpret_cast<float4*>(dx + (bc * H + h) * (int64_t)W + w) = o;
}

// out[c] = sum_{b,hw} x[b,c,hw]: grid (C, S) fixed-order fp64 partials, then one thread per channel sums them
// `direct` != NULL (single split): the channel's sum goes straight to the output -- one launch instead of two
__global__ __launch_bounds__(256) void channel_sum_partial_kernel(const float* __restrict__ x, double* __restrict__ ws, int B, int C, int HW,
                                                                  float* __restrict__ direct) {
  const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
  const int beg = (int)((int64_t)HW * s / S), end = (int)((int64_t)HW * (s + 1) / S);
  float part = 0.f;
  for (int b = 0; b < B; ++b) {
    const float* p = x + ((size_t)b * C + c) * HW;
    for (int i = beg + threadIdx.x; i < end; i += 256) part += p[i];
  }
  double d = wave_sum_d((double)part);
  __shared__ double sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = sm[0] + sm[1] + sm[2] + sm[3];
    if (direct) direct[c] = (float)t; else ws[(size_t)c * S + s] = t;
  }
}

__global__ void channel_sum_final_kernel(const double* __restrict__ ws, float* __restrict__ out, int C, int S) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double t = 0.0;
  for (int s = 0; s < S; ++s) t += ws[(size_t)c * S + s];
  out[c] = (float)t;
}

struct FwdPlan { int tm, tn, bk, splits, wm, wn; };   // block tile = (32*wm*tm) x (32*wn*tn), 64*wm*wn threads

// Tail split (conv_igemm_kernel): for an unsplit launch of `tiles` workgroups, how many of the last tiles to cut along K and
// into how many pieces.  Residency: 4 workgroups per CU for the 128-row / 128-column tiles, 8 for 64 x 64.  Returns the number
// of tail tiles (0 = no tail split) and sets *pieces.  PRN_CONV_TAIL=0 switches it off (A/B).
int plan_tail(const FwdPlan& p, int64_t tiles, int tilesM, int K, int64_t N, int HoWo, int* pieces) {
  static const int on = prn_env_int("PRN_CONV_TAIL", 1);
  *pieces = 0;
  if (!on || p.splits != 1 || p.wm != 2 || (HoWo & 3) != 0) return 0;
  const int64_t slots = 256 * ((p.tm * p.tn >= 2) ? 4 : 8);
  const int64_t full = tiles / slots * slots, r = tiles - full;
  const int kt = cdiv(K, 16);
  // deep-K launches only: on the short-K 1x1 layers (16 .. 64 slices) the pieces are all prologue and epilogue -- 256->1024 @30x40
  // went from 50 to 57 us with its 176 tail tiles cut in five, while the 3x3 layers with K >= 1152 gain 5-6 %
  if (full == 0 || r == 0 || 20 * r < slots || 10 * r > 6 * slots || r % tilesM != 0 || kt < 64) return 0;     // 5 % .. 60 % of a round
  if ((N - (int64_t)(full / tilesM) * 32 * p.wn * p.tn) % 4 != 0) return 0;
  int s = (int)(slots / r);
  if (s > kt / 4) s = kt / 4;
  if (s > 8) s = 8;
  if (s < 2) return 0;
  *pieces = s;
  return (int)r;
}

// All workgroups of these launches are resident at once, so the launch lasts as long as its most loaded CU: 528 workgroups
// (16 CUs with 3, the rest with 2) ran 27 % longer than 512.  Round the split count down so that tiles * splits lands on
// (just below) a multiple of the 256 CUs when only a few workgroups per CU exist (never below 3/4 of the request).
int quantise_splits(int64_t tiles, int splits) {
  if (splits <= 1 || tiles * splits >= 256 * 12) return splits;
  int best = splits;
  double best_eff = 0.0;
  for (int s = splits; s >= 1 && 4 * s > 3 * splits; --s) {       // candidates in (3/4 * splits, splits], ties -> larger
    const double q = (double)(tiles * s) / 256.0;
    const double eff = q / (double)((tiles * s + 255) / 256);
    if (eff > best_eff + 1e-9) { best_eff = eff; best = s; }
  }
  return best;
}

// Tile / slice / split choice.  PRN_CONV_FORCE="tm,tn,bk,splits" overrides it (tuning sweeps: tools/conv_bench.py).
FwdPlan plan_fwd(int M, int64_t N, int K, bool narrow_ok = false, int phases = 1, bool nosplit = false, int ks = 0) {
  static const prn_env4 forced_ = prn_env_ints("PRN_CONV_FORCE");
  const int* forced = forced_.v;
  FwdPlan p;
  p.wm = 2; p.wn = 2;
  if (forced[0] > 0) {
    p.tm = forced[0]; p.tn = forced[1]; p.bk = 16; p.splits = forced[3];
    if (p.tm != p.tn && !(p.tm == 2 && p.tn == 1 && ks != 0)) { p.tm = 1; p.tn = 1; }
  } else {
    // Rule fitted to the sweep of tools/conv_sweep.py over the PlaneRecNet shapes (profiles/r01_conv_sweep.txt, re-run after
    // the buffer-load rewrite): 16-deep slices always; 128x128 tiles only for wide-M, deep-K layers with enough tiles,
    // otherwise 64x64 (64x128 never won a shape); then split K until ~4 workgroups per CU exist (small-N layers).
    auto tiles = [&](int tm, int tn) { return (int64_t)cdiv(M, 64 * tm) * cdiv(N, 64 * tn); };
    p.bk = 16;
    // 128 x 64 tiles (ks = 1 / 3 with zero padding only, else ks == 0): every 3x3 layer with >= 8192 pixels that is too
    // small for 128 x 128 tiles, and the channel-expanding 1x1 layers (M >= 4 C), gain 5-10 % (profiles/r01_conv_sweep_128x64.txt)
    const bool wide = ks != 0 && M >= 128 && cdiv(N, 64) >= 128 && tiles(2, 1) >= 256 && (ks == 3 || M >= 4 * K);
    if (M >= 128 && K >= 1152 && tiles(2, 2) >= 1024) { p.tm = 2; p.tn = 2; }
    else if (wide) { p.tm = 2; p.tn = 1; }
    else if (M >= 128 && K >= 1152 && tiles(2, 2) >= 512) { p.tm = 2; p.tn = 2; }
    else { p.tm = 1; p.tn = 1; }
    int64_t t = tiles(p.tm, p.tn);
    if (M <= 32 && narrow_ok) { p.wm = 1; p.wn = 4; t = cdiv(N, 128); }        // 32 x 128 tile
    t *= phases;
    p.splits = t < 1024 ? (int)((1024 + t - 1) / t) : 1;
    if (p.splits > 8) p.splits = 8;
    p.splits = quantise_splits(t, p.splits);
  }
  const int kt = cdiv(K, p.bk);
  if (p.splits > kt / 4) p.splits = kt / 4;        // at least 4 K slices per split
  if (p.splits > 16) p.splits = 16;
  if (p.splits < 1 || nosplit) p.splits = 1;
  return p;
}

// kernel size if (kernel size, input mode) has the 128 x 64 instance, else 0
constexpr int wide_ks(int ks, int mode) { return (mode == PRN_IN_ZERO && (ks == 1 || ks == 3)) ? ks : 0; }

// (kernel size, input mode) pairs that have the 32 x 128 instance
constexpr bool narrow_available(int ks, int mode) {
  return (ks == 3 && (mode == PRN_IN_ZERO || mode == PRN_IN_REFLECT)) || (ks == 1 && mode == PRN_IN_ZERO);
}

// epilogue through the LDS transpose (float4 stores)?
bool wide_store_ok(const ConvArgs& a, const FwdPlan& p) {
  static const int wide = prn_env_int("PRN_CONV_WIDE_STORE", 1);                                    // PRN_CONV_WIDE_STORE=0: the per-element epilogue everywhere (A/B)
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return wide && a.seg.nseg == 0 && p.wm == 2 && al16(a.y) && al16(a.addend) && al16(a.ws) && (a.zy & 3) == 0 && (a.HoWo & 3) == 0 &&
         (p.splits == 1 || (((int64_t)a.B * a.M * a.HoWo) & 3) == 0);
}

// K-split (or tail-split) sum inside the GEMM launch?  Needs the caller's zeroed tile counters (prn_conv2d_fwd_counted), the
// float4 epilogue, a dense output and partial slices addressable through one buffer descriptor.
// OFF unless PRN_CONV_FUSED_REDUCE=1 (read per call): bit-identical to the two-kernel path and 2-8 % faster per split layer
// in isolation (62 -> 57 us on 1x1 1024->256 @30x40, 118 -> 108 us on 2304->256), but the training step does not move
// (54.8-54.9 ms either way, four alternating runs): inside a step the 7 us sum kernels already ran in the shadow of the
// weight-gradient stream, and the fold lengthens conv_igemm_kernel itself (roofline.frac 0.54 -> 0.53).
int conv_ilv() {
  static const int ilv = prn_env_int("PRN_CONV_ILV", 3);                                     // PRN_CONV_ILV=0: the phase-separated K loops (A/B runs)
  return ilv;
}

bool fused_reduce_ok(const ConvArgs& a, const FwdPlan& p, int phases, int tail_pieces, const unsigned* counters) {
  if (counters == nullptr) return false;
  static const int on = prn_env_int("PRN_CONV_FUSED_REDUCE", 0);
  const int pieces = p.splits > 1 ? p.splits : tail_pieces;
  const int64_t tiles = (int64_t)cdiv(a.M, 32 * p.wm * p.tm) * cdiv(a.N, 32 * p.wn * p.tn);
  return on && counters != nullptr && pieces > 1 && phases == 1 && a.ystride == 1 && a.zy == 0 && tiles <= PRN_TILE_COUNTERS && wide_store_ok(a, p) &&
         (int64_t)pieces * a.B * a.M * a.HoWo * 4 < (1LL << 31);
}

template <int KS, int MODE>
int launch_fwd(const ConvArgs& a0, const FwdPlan& p, hipStream_t st, int phases = 1, int tail_tiles = 0, int tail_pieces = 0, unsigned* counters = nullptr) {
  ConvArgs a = a0;
  a.tilesM = cdiv(a.M, 32 * p.wm * p.tm);
  a.nblocks = a.tilesM * cdiv(a.N, 32 * p.wn * p.tn);
  a.splits = p.splits;
  a.wide_store = wide_store_ok(a, p);
  a.ilv = conv_ilv() & 1;                                  // PRN_CONV_ILV: bit 0 = forward / input-gradient kernel, bit 1 = weight-gradient kernel
  a.tail_first = a.nblocks; a.tail_splits = 0;
  int gx = a.nblocks;
  if (tail_pieces > 1 && phases == 1 && a.seg.nseg == 0 && a.ystride == 1) {
    a.tail_first = a.nblocks - tail_tiles; a.tail_splits = tail_pieces;
    gx = a.tail_first + tail_tiles * tail_pieces;
  }
  a.cnt = fused_reduce_ok(a, p, phases, a.tail_splits, counters) ? counters : nullptr;
  dim3 grid(gx, p.splits, phases), block(64 * p.wm * p.wn);
  if constexpr (narrow_available(KS, MODE)) {
    if (p.wm == 1) {
      hipLaunchKernelGGL((conv_igemm_kernel<KS, MODE, 1, 1, 16, false, 1>), grid, block, 0, st, a);
      return 0;
    }
  }
  if constexpr (KS == 1 && MODE == PRN_IN_ZERO) {
    const bool vec = a.seg.nseg == 0 && a.stride == 1 && a.pad == 0 && (a.HW & 3) == 0 && a.HW == a.HoWo && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
    if (vec) {
      if (p.tm == 2 && p.tn == 2) hipLaunchKernelGGL((conv_igemm_kernel<1, PRN_IN_ZERO, 2, 2, 16, true>), grid, block, 0, st, a);
      else if (p.tm == 2 && p.tn == 1) hipLaunchKernelGGL((conv_igemm_kernel<1, PRN_IN_ZERO, 2, 1, 16, true>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((conv_igemm_kernel<1, PRN_IN_ZERO, 1, 1, 16, true>), grid, block, 0, st, a);
      return 0;
    }
  }
#define PRN_LAUNCH(TM_, TN_, BK_) hipLaunchKernelGGL((conv_igemm_kernel<KS, MODE, TM_, TN_, BK_>), grid, block, 0, st, a)
  // Measured and NOT instantiated (profiles/r01_conv_sweep_bufferloads.txt and the sweeps after it): 32-deep K slices
  // (slower or equal on every shape), 64 x 128 tiles (never the best), and a two-wave 64 x 32 tile meant to replace the
  // K split on the 9600-pixel stages (64 -> 71 us on 1x1 1024->256, 144 -> 193 us on 3x3 256: the split is cheaper).
  if constexpr (KS == 3 && MODE == PRN_IN_ZERO) {
    static const int v3on = prn_env_int("PRN_CONV_V3", 1);                                  // PRN_CONV_V3=0 switches the float4 gather off (A/B)
    bool v3 = v3on && a.stride == 1 && a.pad == 1 && p.wm == 2 && p.wn == 2;
    if (a.seg.nseg > 0) { for (int sI = 0; sI < a.seg.nseg; ++sI) v3 = v3 && (a.seg.W[sI] & 3) == 0; }
    else v3 = v3 && (a.W & 3) == 0 && a.Wo == a.W && a.Ho == a.H;
    if (v3) {
#define PRN_LAUNCH_V3(TM_, TN_) hipLaunchKernelGGL((conv_igemm_kernel<3, PRN_IN_ZERO, TM_, TN_, 16, false, 2, 2, true>), grid, block, 0, st, a)
      if (p.tm == 2 && p.tn == 2) PRN_LAUNCH_V3(2, 2); else if (p.tm == 2 && p.tn == 1) PRN_LAUNCH_V3(2, 1); else PRN_LAUNCH_V3(1, 1);
#undef PRN_LAUNCH_V3
      return 0;
    }
  }
  if constexpr (wide_ks(KS, MODE) != 0) {
    if (p.tm == 2 && p.tn == 1) { PRN_LAUNCH(2, 1, 16); return 0; }
  }
  if (p.tm == 2 && p.tn == 2) PRN_LAUNCH(2, 2, 16);
  else PRN_LAUNCH(1, 1, 16);
#undef PRN_LAUNCH
  return 0;
}

struct WgPlan { int tm, tj, tilesM, tilesJ, splits, chunks, wm; };
WgPlan plan_wgrad(const prn_gemm_opts& o, int M, int K, int64_t N, int phases = 1, bool ragged = false, bool small = false) {
  WgPlan p;
  p.tm = (M > 64) ? 2 : 1;
  p.tj = (K > 64) ? 2 : 1;
  p.wm = 2;
  static const prn_env4 ft_ = prn_env_ints("PRN_WGRAD_TILE");            // PRN_WGRAD_TILE="tm,tj" overrides the tile (tuning sweeps)
  const int ftm = ft_.v[0], ftj = ft_.v[1];
  if (ftm > 0) { p.tm = M > 32 ? ftm : 1; p.tj = K > 32 ? ftj : 1; }
  if (small) { p.tm = 1; p.tj = 1; }
  if (M <= 32 && K > 64) { p.wm = 1; p.tm = 1; p.tj = 1; }          // 32 x 128 tile
  if (ragged) {                                                      // ragged batches: the two instantiated tiles only
    if (M <= 32) { p.wm = 1; p.tm = 1; p.tj = 1; } else { p.wm = 2; p.tm = 2; p.tj = 2; }
  }
  p.tilesM = cdiv(M, 32 * p.wm * p.tm);
  p.tilesJ = cdiv(K, 32 * (4 / p.wm) * p.tj);
  p.chunks = cdiv(N, 16);
  const int tiles = p.tilesM * p.tilesJ * phases;
  // Splits.  Every workgroup of a wgrad launch is resident at once when tiles * splits <= 256 CUs * R (R = workgroups a
  // CU holds: 3 for the 128x128 tile at 152 VGPRs, 4 / 6 for the smaller ones), and the launch then lasts as long as its
  // most loaded CU.  The split sweep (profiles/r01_wgrad_split_sweep.txt) has its minimum where that single round is
  // exactly full (e.g. 36 tiles x 21 splits = 756 of 768 slots: 131 us, vs 156 us at 28 splits and 154 us at 33), so:
  // fill one round; when the bounds below cut that short, land on a whole number of workgroups per CU instead.
  // Bounds: the workspace round trip (2 * splits * M*K*4 bytes) stays a fraction of the MFMA time (splits <= 0.0035 *
  // pixels) and every split keeps at least 128 pixels.  Layers with more tiles than slots only split to ~2048 workgroups.
  const int target = o.wgrad_target > 0 ? o.wgrad_target : 2048;   // (opts: planerecnet_amd.ops lowers it while weight gradients are deferred, like wgrad_wgs)
  static const int forced = prn_env_int("PRN_WGRAD_SPLITS", 0);          // PRN_WGRAD_SPLITS forces the split count (tuning sweeps)
  int R = (p.tm == 2 && p.tj == 2) ? 3 : ((p.tm + p.tj == 3 || p.wm == 1) ? 4 : 6);
  // opts.wgrad_wgs (planerecnet_amd.ops sets it while weight gradients are deferred to the side stream): plan a
  // launch whose tiles are fewer for THIS many workgroups instead of a full residency round.  A weight gradient that shares the GPU
  // with the main chain should not fill every CU's registers (3 resident workgroups of 152 VGPRs leave no room for a 128-VGPR wave
  // of the main stream's GEMMs): 512 workgroups per launch gave the shortest training step (52.6 -> 52.2 ms; 448: 52.3, 640: 52.4,
  // 768: 52.45, 1024: 52.8), although the launch alone is fastest with the full round.
  const int wtot = o.wgrad_wgs;
  int slots = 256 * R;
  if (wtot > 0) { slots = wtot; R = 3; }
  const int sbw = (int)(0.0035 * (double)N) > 1 ? (int)(0.0035 * (double)N) : 1;
  const int smax = p.chunks / 8 > 0 ? p.chunks / 8 : 1;   // at least 128 pixels per split
  int cap = sbw < smax ? sbw : smax;
  if (cap > 256) cap = 256;
  int s;
  if (tiles < slots) {
    s = (R == 3 ? 1 : 2) * slots / tiles;       // the smaller tiles ran ~3 % faster with two rounds (deconv4: 2198 -> 2141 us)
    if (s > cap) s = quantise_splits(tiles, cap);
  } else {
    s = target / tiles;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    s = quantise_splits(tiles, s);
  }
  if (s < 1) s = 1;
  if (forced > 0) s = forced < smax ? forced : smax;
  p.splits = s;
  return p;
}

template <int KS, int MODE>
int launch_wgrad(WgArgs a, const WgPlan& p, hipStream_t st, int phases = 1) {
  a.nzg = phases;
  a.ilv = conv_ilv() & 2 ? 1 : 0;
  dim3 grid((unsigned)(p.tilesM * p.tilesJ * p.splits * phases)), block(256);
  if constexpr ((KS == 1 || KS == 3) && MODE == PRN_IN_ZERO) {
    if (a.seg.nseg > 0) {                                  // plan_wgrad(..., ragged) only hands out these two tiles
      if (p.wm == 1) hipLaunchKernelGGL((conv_wgrad_kernel<KS, MODE, 1, 1, 1, true>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((conv_wgrad_kernel<KS, MODE, 2, 2, 2, true>), grid, block, 0, st, a);
      return 0;
    }
  }
  if (p.wm == 1) hipLaunchKernelGGL((conv_wgrad_kernel<KS, MODE, 1, 1, 1>), grid, block, 0, st, a);
  else if (p.tm == 2 && p.tj == 2) hipLaunchKernelGGL((conv_wgrad_kernel<KS, MODE, 2, 2>), grid, block, 0, st, a);
  else if (p.tm == 2) hipLaunchKernelGGL((conv_wgrad_kernel<KS, MODE, 2, 1>), grid, block, 0, st, a);
  else if (p.tj == 2) hipLaunchKernelGGL((conv_wgrad_kernel<KS, MODE, 1, 2>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((conv_wgrad_kernel<KS, MODE, 1, 1>), grid, block, 0, st, a);
  return 0;
}

int check_desc(const prn_conv_desc* d, const char* who) {
  PRN_REQUIRE(d != nullptr, "%s: null descriptor", who);
  PRN_REQUIRE(d->KH == d->KW && (d->KH == 1 || d->KH == 2 || d->KH == 3 || d->KH == 4 || d->KH == 7), "%s: kernel %dx%d unsupported (1,2,3,4,7 square)", who,
              d->KH, d->KW);
  PRN_REQUIRE(d->B > 0 && d->C > 0 && d->H > 0 && d->W > 0 && d->M > 0 && d->Ho > 0 && d->Wo > 0, "%s: empty dimension", who);
  PRN_REQUIRE(d->in_mode >= 0 && d->in_mode <= 4, "%s: bad in_mode %d", who, d->in_mode);
  PRN_REQUIRE((d->in_mode == PRN_IN_UP2_PHASE) == (d->KH == 2), "%s: 2x2 kernels are the sub-pixel phases of PRN_IN_UP2_PHASE (and only that)", who);
  PRN_REQUIRE(d->in_mode != PRN_IN_UP2_PHASE || (d->Ho == 2 * d->H && d->Wo == 2 * d->W && d->stride == 1 && d->ystride <= 1),
              "%s: PRN_IN_UP2_PHASE maps [H,W] to [2H,2W]", who);
  PRN_REQUIRE(d->KH != 4 || d->in_mode == PRN_IN_ZERO, "%s: 4x4 kernels only with zero padding", who);
  PRN_REQUIRE(d->ystride == 0 || d->ystride == 1 || (d->ystride == 2 && d->yH >= 2 * d->Ho - 1 && d->yW >= 2 * d->Wo - 1),
              "%s: ystride 2 needs a [yH,yW] plane that holds the strided output", who);
  PRN_REQUIRE(d->in_mode != PRN_IN_DILATED || d->dil == 2, "%s: only dilation 2 is implemented", who);
  PRN_REQUIRE((d->in_mode != PRN_IN_REFLECT && d->in_mode != PRN_IN_UP2_REFLECT) || (d->pad == 1 && d->stride == 1 && d->KH == 3),
              "%s: reflect modes need a 3x3 stride-1 pad-1 conv", who);
  PRN_REQUIRE(d->KH != 7 || d->in_mode == PRN_IN_ZERO || d->in_mode == PRN_IN_DILATED, "%s: 7x7 kernels only with zero padding", who);
  // operands are addressed through buffer descriptors with 31-bit byte offsets
  PRN_REQUIRE((int64_t)d->B * d->C * d->H * d->W < (1LL << 29) && (int64_t)d->B * d->M * d->Ho * d->Wo < (1LL << 29) &&
              ((int64_t)d->M + 128) * d->C * d->KH * d->KW < (1LL << 29) && (int64_t)d->B * d->M * (d->yH > 0 ? d->yH : 1) * (d->yW > 0 ? d->yW : 1) < (1LL << 29),
              "%s: tensor larger than 2 GiB", who);
  return 0;
}

// direct (non-GEMM) paths, see conv3x3_small_m_kernel
bool direct_small_m(const prn_conv_desc* d) {
  return d->KH == 3 && d->stride == 1 && d->pad == 1 && d->M <= 2 && d->C <= 112 && (d->in_mode == PRN_IN_ZERO || d->in_mode == PRN_IN_REFLECT) &&
         d->ystride <= 1 && d->Ho == d->H && d->Wo == d->W && (int64_t)d->B * d->H * d->W >= 65536;
}
bool direct_one_c(const prn_conv_desc* d) {
  return d->KH == 3 && d->stride == 1 && d->C == 1 && d->in_mode == PRN_IN_ZERO && d->ystride <= 1 && (int64_t)d->B * d->Ho * d->Wo >= 65536;
}
int direct_wgrad_splits(const prn_conv_desc* d) {
  int s = 2048 / (d->C * d->M);
  return s < 1 ? 1 : (s > 64 ? 64 : s);
}

// GEMM-side geometry of a descriptor: the pixel grid the N dimension runs over, phases, and whether K may be split
struct Geo { int gH, gW, phases; bool nosplit; };
Geo geo_of(const prn_conv_desc* d) {
  Geo g;
  g.phases = d->in_mode == PRN_IN_UP2_PHASE ? 4 : 1;
  g.gH = g.phases == 4 ? d->H : d->Ho;
  g.gW = g.phases == 4 ? d->W : d->Wo;
  g.nosplit = g.phases == 4 || d->ystride == 2;      // the split workspace holds dense [B,M,Ho,Wo] partials only
  return g;
}

// tail split of a dense, unsplit forward launch (plan_tail): number of tail tiles, *pieces = K pieces per tail tile
int tail_of(const prn_conv_desc* d, const Geo& g, const FwdPlan& p, int* pieces) {
  *pieces = 0;
  if (g.phases != 1 || g.nosplit || d->ystride > 1) return 0;
  const int64_t N = (int64_t)d->B * g.gH * g.gW;
  const int tilesM = cdiv(d->M, 32 * p.wm * p.tm);
  return plan_tail(p, (int64_t)tilesM * cdiv(N, 32 * p.wn * p.tn), tilesM, d->C * d->KH * d->KW, N, g.gH * g.gW, pieces);
}

// Plain-GEMM 1x1 convolutions may run on the bf16-split kernel (prn_gemm_split.hip): number of K splits, 0 = no
int split_plan_of(const prn_conv_desc* d) {
  if (!(d->KH == 1 && d->in_mode == PRN_IN_ZERO && d->stride == 1 && d->pad == 0 && d->ystride <= 1 && d->Ho == d->H && d->Wo == d->W)) return 0;
  return prn_split_gemm_plan(d->M, d->C, d->B, d->H * d->W, 1, &d->opts);
}
// Dense stride-1 1x1 weight gradients may run on the 16-bit pipe (prn_wgrad16.hip): pixel splits, 0 = no.  G: layers in the launch.
int wgrad16_plan_of(const prn_conv_desc* d, int G) {
  if (!(d->KH == 1 && d->in_mode == PRN_IN_ZERO && d->stride == 1 && d->pad == 0 && d->ystride <= 1 && d->Ho == d->H && d->Wo == d->W)) return 0;
  return prn_wgrad16_plan(d->M, d->C, (int64_t)d->B * d->H * d->W, d->H * d->W, G, &d->opts);
}
// 4x4 / stride-2 zero-padded convolutions (the input gradient of the sub-pixel upsample-convolutions) on the 16-bit pipe by tap gather
// (prn_split_conv_taps): K splits, 0 = keep the fp32 implicit GEMM
int taps_plan_of(const prn_conv_desc* d) {
  static const int on = prn_env_int("PRN_SPLIT_TAPS", 1);                                      // PRN_SPLIT_TAPS=0: off (A/B)
  // (also the stride-2 1x1 downsample convolutions of the backbone's stage entries, models/backbone.py:45: one tap, every second pixel)
  const bool shape = (d->KH == 4 && d->stride == 2) || (d->KH == 1 && d->stride == 2 && d->pad == 0);
  if (!on || !(shape && d->in_mode == PRN_IN_ZERO && d->ystride <= 1 && (d->C & 31) == 0 && d->opts.split_kind == PRN_PIECES_F16)) return 0;
  return prn_split_gemm_plan(d->M, d->C * d->KH * d->KW, d->B, d->Ho * d->Wo, 1, &d->opts);
}
// the sub-pixel phases themselves (forward of the upsample-convolutions): 1 = on the 16-bit pipe (four phases as the z axis of one launch), 0 = fp32
int up2_plan_of(const prn_conv_desc* d) {
  static const int on = prn_env_int("PRN_SPLIT_UP2", 1);                                      // PRN_SPLIT_UP2=0: off (A/B)
  if (!on || !(d->KH == 2 && d->in_mode == PRN_IN_UP2_PHASE && (d->C & 31) == 0 && (d->W & 3) == 0 && d->opts.split_kind == PRN_PIECES_F16 && d->ystride <= 1)) return 0;
  if (!(d->epilogue == PRN_EPI_NONE || d->epilogue == PRN_EPI_RELU)) return 0;               // (the phase launch's epilogue has bias and ReLU only: a sigmoid stays on the fp32 kernel)
  return prn_split_gemm_plan(d->M, d->C * 4, d->B, d->H * d->W, 4, &d->opts) == 1 ? 1 : 0;
}
int64_t taps_ws_bytes(const prn_conv_desc* d, int splits) {
  return ((prn_split_gemm_image_bytes(d->M, d->C * d->KH * d->KW, 1) + 255) & ~255LL) + prn_split_gemm_partial_bytes(d->M, d->B, d->Ho * d->Wo, 1, splits);
}
int64_t split_ws_bytes(const prn_conv_desc* d, int splits) {
  return ((prn_split_gemm_image_bytes(d->M, d->C, 1) + 255) & ~255LL) + prn_split_gemm_partial_bytes(d->M, d->B, d->H * d->W, 1, splits);
}

}  // namespace

int prn_launch_reduce_epilogue(const float* ws, const float* bias, const float* addend, float* y, int64_t total, int M, int HoWo, int splits,
                               int epi, hipStream_t st) {
  hipLaunchKernelGGL(reduce_epilogue_kernel, dim3(cdiv(total, 1024)), dim3(256), 0, st, ws, bias, addend, y, total, M, HoWo, splits, epi);
  PRN_CHECK_LAUNCH("reduce_epilogue");
  return 0;
}
int prn_launch_reduce_splits(const float* ws, float* out, int64_t n, int splits, hipStream_t st) {
  for (int r = PRN_REPS(4); r > 0; --r) hipLaunchKernelGGL(reduce_splits_kernel, dim3(cdiv(n, 64)), dim3(256), 0, st, ws, out, n, splits);
  PRN_CHECK_LAUNCH("reduce_splits");
  return 0;
}
int prn_quantise_splits(int64_t tiles, int splits) { return quantise_splits(tiles, splits); }

extern "C" int64_t prn_conv2d_fwd_ws_bytes(const prn_conv_desc* d) {
  if (check_desc(d, "prn_conv2d_fwd_ws_bytes")) return -1;
  if (direct_small_m(d) || direct_one_c(d)) return 0;
  if (const int ss = split_plan_of(d)) return split_ws_bytes(d, ss);
  if (const int st = taps_plan_of(d)) return taps_ws_bytes(d, st);
  if (up2_plan_of(d)) return (prn_split_gemm_image_bytes(d->M, d->C * 4, 4) + 255) & ~255LL;
  const Geo g = geo_of(d);
  const FwdPlan p = plan_fwd(d->M, (int64_t)d->B * g.gH * g.gW, d->C * d->KH * d->KW, narrow_available(d->KH, d->in_mode), g.phases, g.nosplit,
                             wide_ks(d->KH, d->in_mode));
  if (p.splits > 1) return (int64_t)p.splits * d->B * d->M * d->Ho * d->Wo * 4;
  int pieces = 0;
  tail_of(d, g, p, &pieces);
  return pieces > 1 ? (int64_t)pieces * d->B * d->M * d->Ho * d->Wo * 4 : 0;
}

extern "C" int prn_conv2d_fwd(const prn_conv_desc* d, const float* x, const float* w, const float* bias,
                              const float* addend, float* y, void* ws, void* stream) {
  return prn_conv2d_fwd_phase(d, x, w, bias, addend, y, ws, stream, 0);
}

namespace {
// Segment table of a ragged batch (nullptr / nseg 0: dense).  `align` = pixels per tile / chunk that must not straddle segments.
int fill_seg(Seg& sg, const prn_ragged* rg, const prn_conv_desc* d, int align, const char* who) {
  sg.nseg = 0;
  if (rg == nullptr) return 0;
  PRN_REQUIRE(rg->nseg >= 1 && rg->nseg <= MAX_SEG, "%s: 1..%d segments", who, MAX_SEG);
  PRN_REQUIRE(d->in_mode == PRN_IN_ZERO && d->stride == 1 && (d->KH == 1 || d->KH == 3) && d->pad == (d->KH - 1) / 2 && d->ystride <= 1,
              "%s: ragged batches take stride-1 'same' 1x1 / 3x3 convolutions with zero padding", who);
  int64_t n = 0, xo = 0, yo = 0;
  for (int s = 0; s < rg->nseg; ++s) {
    const int H = rg->H[s], W = rg->W[s];
    PRN_REQUIRE(H > 0 && W > 0, "%s: empty segment", who);
    PRN_REQUIRE(n % align == 0, "%s: segment %d starts at pixel %lld, not a multiple of %d", who, s, (long long)n, align);
    sg.n0[s] = (int)n; sg.H[s] = H; sg.W[s] = W; sg.x0[s] = (int)xo; sg.y0[s] = (int)yo;
    n += (int64_t)d->B * H * W;
    xo += (int64_t)d->B * d->C * H * W;
    yo += (int64_t)d->B * d->M * H * W;
  }
  PRN_REQUIRE(n % align == 0 && n < (1LL << 31) && xo < (1LL << 29) && yo < (1LL << 29), "%s: ragged batch too large or unaligned", who);
  for (int s = rg->nseg; s <= MAX_SEG; ++s) sg.n0[s] = (int)n;
  sg.nseg = rg->nseg;
  return 0;
}
int64_t seg_pixels(const prn_ragged* rg, int B) {
  int64_t n = 0;
  for (int s = 0; s < rg->nseg; ++s) n += (int64_t)B * rg->H[s] * rg->W[s];
  return n;
}
int conv_fwd_impl(const prn_conv_desc* d, const prn_ragged* rg, const float* x, const float* w, const float* bias, const float* addend, float* y,
                  void* ws, void* stream, int phase, unsigned* counters = nullptr, const void* w_images = nullptr);
int conv_wgrad_impl(const prn_conv_desc* d, const prn_ragged* rg, const float* x, const float* dy, float* dw, void* ws, void* stream, int phase);
}  // namespace

// which kernel family a descriptor's forward / weight gradient runs on: 0 = the MFMA implicit GEMM, 1 = the direct (one thread per
// pixel, HBM-bound) kernels for one- or two-channel 3x3 layers -- so that a profiler can attribute a launch to the right roofline
extern "C" int prn_conv2d_kernel_kind(const prn_conv_desc* d) {
  if (check_desc(d, "prn_conv2d_kernel_kind")) return -1;
  if (direct_small_m(d) || direct_one_c(d)) return 1;
  int ss = split_plan_of(d);
  if (ss == 0) ss = taps_plan_of(d);
  if (ss == 0) ss = up2_plan_of(d);
  return ss == 0 ? 0 : (ss == 1 ? 2 : 3);
}

// Where phase 1 of the descriptor's forward leaves its K-split partial sums: returns s > 1 when ws + *offset_bytes holds s dense [B][M][Ho][Wo]
// tensors (B*M*Ho*Wo elements apart) whose sum in split order, + bias + addend, is the result (what phase 2 computes); 0 when the launch has no
// K split, or splits only its tail tiles.  A consumer that reads the operator's output exactly once may sum the partials itself instead of
// running phase 2 (prn_bn_train_fwd_partials / prn_bn_bwd_partials); phase 1 is then called with counters == NULL (no in-launch sum).
extern "C" int prn_conv2d_fwd_partials(const prn_conv_desc* d, int64_t* offset_bytes) {
  if (check_desc(d, "prn_conv2d_fwd_partials")) return -1;
  PRN_REQUIRE(offset_bytes != nullptr, "prn_conv2d_fwd_partials: null offset");
  *offset_bytes = 0;
  if (direct_small_m(d) || direct_one_c(d)) return 0;
  if (const int ss = split_plan_of(d)) {
    *offset_bytes = (prn_split_gemm_image_bytes(d->M, d->C, 1) + 255) & ~255LL;
    return ss > 1 ? ss : 0;
  }
  if (taps_plan_of(d) || up2_plan_of(d)) return 0;
  const Geo g = geo_of(d);
  const FwdPlan p = plan_fwd(d->M, (int64_t)d->B * g.gH * g.gW, d->C * d->KH * d->KW, narrow_available(d->KH, d->in_mode), g.phases, g.nosplit,
                             wide_ks(d->KH, d->in_mode));
  return p.splits > 1 ? p.splits : 0;
}

extern "C" int prn_conv2d_fwd_phase(const prn_conv_desc* d, const float* x, const float* w, const float* bias,
                                    const float* addend, float* y, void* ws, void* stream, int phase) {
  return conv_fwd_impl(d, nullptr, x, w, bias, addend, y, ws, stream, phase);
}

extern "C" int prn_conv2d_fwd_counted(const prn_conv_desc* d, const float* x, const float* w, const void* w_images, const float* bias, const float* addend,
                                      float* y, void* ws, unsigned* counters, void* stream, int phase) {
  return conv_fwd_impl(d, nullptr, x, w, bias, addend, y, ws, stream, phase, counters, w_images);
}

extern "C" int prn_conv2d_fwd_ragged(const prn_conv_desc* d, const prn_ragged* rg, const float* x, const float* w, const float* bias,
                                     const float* addend, float* y, void* stream) {
  PRN_REQUIRE(rg != nullptr, "prn_conv2d_fwd_ragged: null segment list");
  return conv_fwd_impl(d, rg, x, w, bias, addend, y, nullptr, stream, 0);
}

namespace {
int conv_fwd_impl(const prn_conv_desc* d0, const prn_ragged* rg, const float* x, const float* w, const float* bias, const float* addend, float* y,
                  void* ws, void* stream, int phase, unsigned* counters, const void* w_images) {
  prn_conv_desc dd;
  const prn_conv_desc* d = d0;
  if (rg && d0) {                                       // geometry fields of the descriptor are per segment: validate with the first one
    dd = *d0; dd.H = dd.Ho = rg->H[0]; dd.W = dd.Wo = rg->W[0]; d = &dd;
  }
  if (int e = check_desc(d, "prn_conv2d_fwd")) return e;
  PRN_REQUIRE(x && w && y, "prn_conv2d_fwd: null tensor");
  if (rg == nullptr && (direct_small_m(d) || direct_one_c(d))) {
    if (phase == 2) return 0;
    hipStream_t sd = (hipStream_t)stream;
    if (direct_small_m(d)) {
      const dim3 grid(cdiv((int64_t)d->B * d->H * d->W, 256)), block(256);
#define PRN_SMALL_M(MODE_, MM_) hipLaunchKernelGGL((conv3x3_small_m_kernel<MODE_, MM_>), grid, block, 0, sd, x, w, bias, addend, y, d->B, d->C, d->H, d->W, \
                                                   d->pad, d->epilogue)
      if (d->in_mode == PRN_IN_ZERO) { if (d->M == 1) PRN_SMALL_M(PRN_IN_ZERO, 1); else PRN_SMALL_M(PRN_IN_ZERO, 2); }
      else { if (d->M == 1) PRN_SMALL_M(PRN_IN_REFLECT, 1); else PRN_SMALL_M(PRN_IN_REFLECT, 2); }
#undef PRN_SMALL_M
    } else {
      hipLaunchKernelGGL(conv3x3_one_c_kernel, dim3(cdiv((int64_t)d->B * d->Ho * d->Wo, 256)), dim3(256), 0, sd, x, w, bias, addend, y, d->B, d->H, d->W,
                         d->M, d->pad, d->Ho, d->Wo, d->epilogue);
    }
    PRN_CHECK_LAUNCH("prn_conv2d_fwd/direct");
    return 0;
  }
  if (rg == nullptr && ws != nullptr) {
    if (const int ss = split_plan_of(d)) {
      const int64_t ib = (prn_split_gemm_image_bytes(d->M, d->C, 1) + 255) & ~255LL;
      return prn_split_gemm(w, w_images, x, bias, addend, y, ws, (float*)((char*)ws + ib), d->M, d->C, d->B, d->H * d->W, 1, 0, 0, 0, d->epilogue, ss, &d->opts,
                            (hipStream_t)stream, phase);
    }
    if (up2_plan_of(d) && addend == nullptr && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
      if (phase == 2) return 0;
      return prn_split_conv_up2(w, x, bias, y, ws, d->M, d->C, d->B, d->H, d->W, d->epilogue, &d->opts, (hipStream_t)stream);
    }
    if (const int st = taps_plan_of(d)) {
      const int64_t ib = (prn_split_gemm_image_bytes(d->M, d->C * d->KH * d->KW, 1) + 255) & ~255LL;
      return prn_split_conv_taps(w, x, bias, addend, y, ws, (float*)((char*)ws + ib), d->M, d->C, d->B, d->H, d->W, d->Ho, d->Wo, d->KH, d->KW, d->stride, d->pad, d->epilogue,
                                 st, &d->opts, (hipStream_t)stream, phase);
    }
  }
  ConvArgs a;
  a.x = x; a.w = w; a.bias = bias; a.addend = addend; a.y = y; a.ws = (float*)ws;
  a.B = d->B; a.C = d->C; a.H = d->H; a.W = d->W; a.M = d->M; a.stride = d->stride; a.pad = d->pad;
  const Geo g = geo_of(d);
  a.Ho = g.gH; a.Wo = g.gW; a.epi = d->epilogue;
  a.K = d->C * d->KH * d->KW; a.N = d->B * g.gH * g.gW; a.HoWo = g.gH * g.gW; a.HW = d->H * d->W;
  a.xbytes = d->B * d->C * d->H * d->W * 4; a.wbytes = d->M * a.K * 4;
  a.ystride = 1; a.yW = g.gW; a.yHW = a.HoWo;
  a.zx = a.zw = a.zy = 0;
  if (g.phases == 4) { a.ystride = 2; a.yW = d->Wo; a.yHW = d->Ho * d->Wo; }
  else if (d->ystride == 2) { a.ystride = 2; a.yW = d->yW; a.yHW = d->yH * d->yW; }
  if (rg) a.N = (int)seg_pixels(rg, d->B);
  FwdPlan p = plan_fwd(a.M, a.N, a.K, narrow_available(d->KH, d->in_mode), g.phases, g.nosplit || rg != nullptr, wide_ks(d->KH, d->in_mode));
  if (rg) {                                             // tiles must not straddle segments: fall back from 128 to 64 pixels per tile
    int bn = 32 * p.wn * p.tn;
    bool ok = true;
    int64_t n = 0;
    for (int sI = 0; sI < rg->nseg; ++sI) { ok = ok && n % bn == 0; n += (int64_t)d->B * rg->H[sI] * rg->W[sI]; }
    if (!(ok && n % bn == 0) && !(p.tm == 1 && p.tn == 1 && p.wm == 2)) { p.tm = 1; p.tn = 1; p.wm = 2; p.wn = 2; }
    if (int e = fill_seg(a.seg, rg, d, 32 * p.wn * p.tn, "prn_conv2d_fwd_ragged")) return e;
  } else {
    a.seg.nseg = 0;
  }
  PRN_REQUIRE(p.splits == 1 || ws != nullptr, "prn_conv2d_fwd: workspace required (%d K-splits, see prn_conv2d_fwd_ws_bytes)", p.splits);
  hipStream_t st = (hipStream_t)stream;
  const int mode = d->in_mode;
  int tail_pieces = 0;
  const int tail_tiles = (rg == nullptr && ws != nullptr) ? tail_of(d, g, p, &tail_pieces) : 0;     // (no workspace handed in: plain launch)
  if (phase == 2) goto reduce_only;
  if (d->KH == 1) {
    PRN_REQUIRE(mode == PRN_IN_ZERO || mode == PRN_IN_DILATED, "prn_conv2d_fwd: 1x1 kernels take zero or dilated input mode");
    if (mode == PRN_IN_ZERO) launch_fwd<1, PRN_IN_ZERO>(a, p, st, 1, tail_tiles, tail_pieces, counters); else launch_fwd<1, PRN_IN_DILATED>(a, p, st, 1, 0, 0, counters);
  } else if (d->KH == 2) {
    launch_fwd<2, PRN_IN_UP2_PHASE>(a, p, st, 4);
  } else if (d->KH == 3) {
    if (mode == PRN_IN_ZERO) launch_fwd<3, PRN_IN_ZERO>(a, p, st, 1, tail_tiles, tail_pieces, counters);
    else if (mode == PRN_IN_REFLECT) launch_fwd<3, PRN_IN_REFLECT>(a, p, st, 1, tail_tiles, tail_pieces, counters);
    else if (mode == PRN_IN_UP2_REFLECT) launch_fwd<3, PRN_IN_UP2_REFLECT>(a, p, st, 1, 0, 0, counters);
    else launch_fwd<3, PRN_IN_DILATED>(a, p, st, 1, 0, 0, counters);
  } else if (d->KH == 4) {
    launch_fwd<4, PRN_IN_ZERO>(a, p, st, 1, 0, 0, counters);
  } else {
    if (mode == PRN_IN_ZERO) launch_fwd<7, PRN_IN_ZERO>(a, p, st, 1, 0, 0, counters); else launch_fwd<7, PRN_IN_DILATED>(a, p, st, 1, 0, 0, counters);
  }
  PRN_CHECK_LAUNCH("prn_conv2d_fwd");
  if (phase == 1) return 0;
reduce_only:
  {
    const bool tail_launch = tail_pieces > 1 && (mode == PRN_IN_ZERO || mode == PRN_IN_REFLECT) && (d->KH == 1 || d->KH == 3);
    if (fused_reduce_ok(a, p, g.phases, tail_launch ? tail_pieces : 0, counters)) return 0;       // summed inside the GEMM launch
  }
  if (p.splits > 1) {
    const int64_t total = (int64_t)a.B * a.M * a.HoWo;
    hipLaunchKernelGGL(reduce_epilogue_kernel, dim3(cdiv(total, 1024)), dim3(256), 0, st, (const float*)ws, bias, addend, y, total, a.M, a.HoWo,
                       p.splits, a.epi);
    PRN_CHECK_LAUNCH("prn_conv2d_fwd/reduce");
  } else if (tail_pieces > 1 && (mode == PRN_IN_ZERO || mode == PRN_IN_REFLECT) && (d->KH == 1 || d->KH == 3)) {
    const int64_t total = (int64_t)a.B * a.M * a.HoWo;
    const int bn = 32 * p.wn * p.tn, tilesM = cdiv(a.M, 32 * p.wm * p.tm);
    const int n_start = (int)(((int64_t)tilesM * cdiv(a.N, bn) - tail_tiles) / tilesM) * bn;
    const int ntail4 = (a.N - n_start) / 4;
    hipLaunchKernelGGL(reduce_tail_kernel, dim3(cdiv((int64_t)a.M * ntail4, 256)), dim3(256), 0, st, (const float*)ws, bias, addend, y, total, a.M, a.HoWo,
                       n_start, ntail4, tail_pieces, a.epi);
    PRN_CHECK_LAUNCH("prn_conv2d_fwd/tail reduce");
  }
  return 0;
}
}  // namespace

extern "C" int64_t prn_conv2d_wgrad_ws_bytes(const prn_conv_desc* d) {
  if (check_desc(d, "prn_conv2d_wgrad_ws_bytes")) return -1;
  const int K = d->C * d->KH * d->KW;
  if (direct_small_m(d)) return (int64_t)direct_wgrad_splits(d) * d->M * K * 4;
  if (const int s16 = wgrad16_plan_of(d, 1)) return s16 > 1 ? (int64_t)s16 * d->M * K * 4 : 0;
  const Geo g = geo_of(d);
  WgPlan p = plan_wgrad(d->opts, d->M, K, (int64_t)d->B * g.gH * g.gW, g.phases);
  return p.splits > 1 ? (int64_t)p.splits * g.phases * d->M * K * 4 : 0;
}

// which kernel a descriptor's weight gradient runs on: 0 = fp32 MFMA (conv_wgrad_kernel), 1 = the direct HBM-bound kernel, 2 = the fp16-piece
// kernel of prn_wgrad16.hip (G = layers per launch, 1 for prn_conv2d_wgrad) -- for profilers that attribute launches to a roofline
extern "C" int prn_conv2d_wgrad_kernel_kind(const prn_conv_desc* d, int G) {
  if (check_desc(d, "prn_conv2d_wgrad_kernel_kind")) return -1;
  if (direct_small_m(d)) return 1;
  return wgrad16_plan_of(d, G < 1 ? 1 : G) ? 2 : 0;
}
extern "C" int prn_conv2d_wgrad(const prn_conv_desc* d, const float* x, const float* dy, float* dw, void* ws, void* stream) {
  return prn_conv2d_wgrad_phase(d, x, dy, dw, ws, stream, 0);
}

extern "C" int prn_conv2d_wgrad_phase(const prn_conv_desc* d, const float* x, const float* dy, float* dw, void* ws, void* stream, int phase) {
  return conv_wgrad_impl(d, nullptr, x, dy, dw, ws, stream, phase);
}

// G layers of one shape in one launch (blockIdx.z = layer): with G x the tiles, the launch fills the CUs with far fewer pixel
// splits than a single small-map layer needs (1x1 1024->256 @30x40: 33 splits alone, 4 with 8 layers), i.e. far less partial
// traffic, and one fixed-order reduction instead of G.
extern "C" int64_t prn_conv2d_wgrad_grouped_ws_bytes(const prn_conv_desc* d, int G) {
  if (check_desc(d, "prn_conv2d_wgrad_grouped_ws_bytes")) return -1;
  if (G < 1 || G > PRN_WGRAD_GROUP_MAX || geo_of(d).phases != 1 || direct_small_m(d)) { prn_set_error("prn_conv2d_wgrad_grouped_ws_bytes: unsupported group"); return -1; }
  const int K = d->C * d->KH * d->KW;
  if (const int s16 = wgrad16_plan_of(d, G)) return s16 > 1 ? (int64_t)s16 * G * d->M * K * 4 : 0;
  const WgPlan p = plan_wgrad(d->opts, d->M, K, (int64_t)d->B * d->Ho * d->Wo, G);
  return p.splits > 1 ? (int64_t)p.splits * G * d->M * K * 4 : 0;
}

extern "C" int prn_conv2d_wgrad_grouped(const prn_conv_desc* d, int G, const float* const* x, const float* const* dy, float* dw, void* ws, void* stream) {
  if (int e = check_desc(d, "prn_conv2d_wgrad_grouped")) return e;
  PRN_REQUIRE(G >= 1 && G <= PRN_WGRAD_GROUP_MAX, "prn_conv2d_wgrad_grouped: 1 <= G <= %d", PRN_WGRAD_GROUP_MAX);
  PRN_REQUIRE(x && dy && dw, "prn_conv2d_wgrad_grouped: null argument");
  const Geo g = geo_of(d);
  PRN_REQUIRE(g.phases == 1 && !direct_small_m(d) && d->in_mode != PRN_IN_DILATED && d->KH != 4 && d->KH != 2 && d->ystride <= 1,
              "prn_conv2d_wgrad_grouped: dense 1x1 / 3x3 / 7x7 weight gradients only");
  if (const int s16 = wgrad16_plan_of(d, G)) {               // both operands cut into fp16 pieces in the launch (prn_wgrad16.hip)
    // The plan (kernel, pixel splits, workspace size: prn_conv2d_wgrad_grouped_ws_bytes) is a function of the descriptor alone, never of
    // the pointers: an operand the planned kernel cannot take is an error, not a silent switch to a kernel with another split count.
    for (int i = 0; i < G; ++i)
      PRN_REQUIRE(x[i] && dy[i] && ((reinterpret_cast<uintptr_t>(x[i]) | reinterpret_cast<uintptr_t>(dy[i])) & 15) == 0,
                  "prn_conv2d_wgrad_grouped: x / dy of layer %d must be non-null and 16-byte aligned (the descriptor plans the 16-bit-pipe kernel; opts.wgrad_split = PRN_SPLIT_OFF selects the fp32 kernel)", i);
    PRN_REQUIRE(s16 == 1 || ws != nullptr, "prn_conv2d_wgrad_grouped: workspace required (%d splits)", s16);
    if (int e = prn_wgrad16_launch(nullptr, nullptr, dy, x, G, s16 > 1 ? (float*)ws : dw, d->M, d->C, d->B, d->H * d->W, G, 0, 0, s16, &d->opts, (hipStream_t)stream)) return e;
    if (s16 > 1) return prn_launch_reduce_splits((const float*)ws, dw, (int64_t)G * d->M * d->C, s16, (hipStream_t)stream);
    return 0;
  }
  WgArgs a;
  a.x = x[0]; a.dy = dy[0];
  a.B = d->B; a.C = d->C; a.H = d->H; a.W = d->W; a.M = d->M; a.stride = d->stride; a.pad = d->pad;
  a.Ho = g.gH; a.Wo = g.gW;
  a.K = d->C * d->KH * d->KW; a.N = d->B * g.gH * g.gW; a.HoWo = g.gH * g.gW; a.HW = d->H * d->W;
  a.xbytes = d->B * d->C * d->H * d->W * 4; a.dybytes = d->B * d->M * a.HoWo * 4;
  a.xz = 0;
  a.seg.nseg = 0;
  a.ngroup = G;
  for (int i = 0; i < PRN_WGRAD_GROUP_MAX; ++i) {
    a.gx[i] = x[i < G ? i : 0]; a.gdy[i] = dy[i < G ? i : 0];
    PRN_REQUIRE(a.gx[i] && a.gdy[i], "prn_conv2d_wgrad_grouped: null tensor in the group");
  }
  const WgPlan p = plan_wgrad(d->opts, a.M, a.K, a.N, G);
  a.tilesM = p.tilesM; a.tilesJ = p.tilesJ; a.splits = p.splits; a.chunks = p.chunks;
  PRN_REQUIRE(p.splits == 1 || ws != nullptr, "prn_conv2d_wgrad_grouped: workspace required (%d splits)", p.splits);
  a.out = p.splits > 1 ? (float*)ws : dw;
  hipStream_t st = (hipStream_t)stream;
  const int mode = d->in_mode;
  if (d->KH == 1) {
    PRN_REQUIRE(mode == PRN_IN_ZERO, "prn_conv2d_wgrad_grouped: 1x1 kernels take zero input mode");
    launch_wgrad<1, PRN_IN_ZERO>(a, p, st, G);
  } else if (d->KH == 3) {
    if (mode == PRN_IN_ZERO) launch_wgrad<3, PRN_IN_ZERO>(a, p, st, G);
    else if (mode == PRN_IN_REFLECT) launch_wgrad<3, PRN_IN_REFLECT>(a, p, st, G);
    else launch_wgrad<3, PRN_IN_UP2_REFLECT>(a, p, st, G);
  } else {
    launch_wgrad<7, PRN_IN_ZERO>(a, p, st, G);
  }
  PRN_CHECK_LAUNCH("prn_conv2d_wgrad_grouped");
  if (p.splits > 1) {
    const int64_t n = (int64_t)G * a.M * a.K;
    for (int r = PRN_REPS(4); r > 0; --r) hipLaunchKernelGGL(reduce_splits_kernel, dim3(cdiv(n, 64)), dim3(256), 0, st, (const float*)ws, dw, n, p.splits);
    PRN_CHECK_LAUNCH("prn_conv2d_wgrad_grouped/reduce");
  }
  return 0;
}

extern "C" int64_t prn_conv2d_wgrad_ragged_ws_bytes(const prn_conv_desc* d, const prn_ragged* rg) {
  if (d == nullptr || rg == nullptr || rg->nseg < 1 || rg->nseg > MAX_SEG) return -1;
  const int K = d->C * d->KH * d->KW;
  WgPlan p = plan_wgrad(d->opts, d->M, K, seg_pixels(rg, d->B), 1, true);
  return p.splits > 1 ? (int64_t)p.splits * d->M * K * 4 : 0;
}

extern "C" int prn_conv2d_wgrad_ragged(const prn_conv_desc* d, const prn_ragged* rg, const float* x, const float* dy, float* dw, void* ws,
                                       void* stream) {
  PRN_REQUIRE(rg != nullptr, "prn_conv2d_wgrad_ragged: null segment list");
  return conv_wgrad_impl(d, rg, x, dy, dw, ws, stream, 0);
}

namespace {
int conv_wgrad_impl(const prn_conv_desc* d0, const prn_ragged* rg, const float* x, const float* dy, float* dw, void* ws, void* stream, int phase) {
  prn_conv_desc dd;
  const prn_conv_desc* d = d0;
  if (rg && d0) { dd = *d0; dd.H = dd.Ho = rg->H[0]; dd.W = dd.Wo = rg->W[0]; d = &dd; }
  if (int e = check_desc(d, "prn_conv2d_wgrad")) return e;
  PRN_REQUIRE(d->in_mode != PRN_IN_DILATED && d->KH != 4 && d->ystride <= 1, "prn_conv2d_wgrad: dgrad-only descriptor (dilated input, 4x4, strided output)");
  PRN_REQUIRE(x && dy && dw, "prn_conv2d_wgrad: null tensor");
  if (rg == nullptr && direct_small_m(d)) {
    PRN_REQUIRE(ws != nullptr, "prn_conv2d_wgrad: workspace required");
    hipStream_t sd = (hipStream_t)stream;
    const int S = direct_wgrad_splits(d);
    const int64_t n = (int64_t)d->M * d->C * 9;
    if (phase != 2) {
      const dim3 grid(d->C * d->M, S), block(256);
      if (d->in_mode == PRN_IN_ZERO) hipLaunchKernelGGL((conv3x3_small_m_wgrad_kernel<PRN_IN_ZERO>), grid, block, 0, sd, x, dy, (float*)ws, d->B, d->C, d->H, d->W, d->M, d->pad);
      else hipLaunchKernelGGL((conv3x3_small_m_wgrad_kernel<PRN_IN_REFLECT>), grid, block, 0, sd, x, dy, (float*)ws, d->B, d->C, d->H, d->W, d->M, d->pad);
      PRN_CHECK_LAUNCH("prn_conv2d_wgrad/direct");
    }
    if (phase != 1) {
      for (int r = PRN_REPS(4); r > 0; --r) hipLaunchKernelGGL(reduce_splits_kernel, dim3(cdiv(n, 64)), dim3(256), 0, sd, (const float*)ws, dw, n, S);
      PRN_CHECK_LAUNCH("prn_conv2d_wgrad/direct reduce");
    }
    return 0;
  }
  if (rg == nullptr) {
    if (const int s16 = wgrad16_plan_of(d, 1)) {             // both operands cut into fp16 pieces in the launch (prn_wgrad16.hip)
      // (plan = f(descriptor) only, as prn_conv2d_wgrad_ws_bytes sized the workspace: a misaligned operand is an error, not a fallback)
      PRN_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0,
                  "prn_conv2d_wgrad: x and dy must be 16-byte aligned (the descriptor plans the 16-bit-pipe kernel; opts.wgrad_split = PRN_SPLIT_OFF selects the fp32 kernel)");
      PRN_REQUIRE(s16 == 1 || ws != nullptr, "prn_conv2d_wgrad: workspace required (%d splits)", s16);
      if (phase != 2)
        if (int e = prn_wgrad16_launch(dy, x, nullptr, nullptr, 0, s16 > 1 ? (float*)ws : dw, d->M, d->C, d->B, d->H * d->W, 1, 0, 0, s16, &d->opts, (hipStream_t)stream)) return e;
      if (phase != 1 && s16 > 1) return prn_launch_reduce_splits((const float*)ws, dw, (int64_t)d->M * d->C, s16, (hipStream_t)stream);
      return 0;
    }
  }
  WgArgs a;
  a.x = x; a.dy = dy;
  a.B = d->B; a.C = d->C; a.H = d->H; a.W = d->W; a.M = d->M; a.stride = d->stride; a.pad = d->pad;
  const Geo g = geo_of(d);
  a.Ho = g.gH; a.Wo = g.gW;
  a.K = d->C * d->KH * d->KW; a.N = d->B * g.gH * g.gW; a.HoWo = g.gH * g.gW; a.HW = d->H * d->W;
  a.xbytes = d->B * d->C * d->H * d->W * 4; a.dybytes = d->B * d->M * a.HoWo * 4;
  a.xz = 0;
  a.ngroup = 0;
  a.seg.nseg = 0;
  if (rg) {
    a.N = (int)seg_pixels(rg, d->B);
    if (int e = fill_seg(a.seg, rg, d, 16, "prn_conv2d_wgrad_ragged")) return e;
    for (int sI = 0; sI < rg->nseg; ++sI) PRN_REQUIRE((rg->H[sI] * rg->W[sI]) % 4 == 0, "prn_conv2d_wgrad_ragged: segment planes must be multiples of 4 pixels");
  }
  WgPlan p = plan_wgrad(d->opts, a.M, a.K, a.N, g.phases, rg != nullptr);
  a.tilesM = p.tilesM; a.tilesJ = p.tilesJ; a.splits = p.splits; a.chunks = p.chunks;
  PRN_REQUIRE(p.splits == 1 || ws != nullptr, "prn_conv2d_wgrad: workspace required (%d splits)", p.splits);
  a.out = p.splits > 1 ? (float*)ws : dw;
  hipStream_t st = (hipStream_t)stream;
  const int mode = d->in_mode;
  if (phase == 2) goto reduce_only;
  if (d->KH == 1) {
    PRN_REQUIRE(mode == PRN_IN_ZERO, "prn_conv2d_wgrad: 1x1 kernels take zero input mode");
    launch_wgrad<1, PRN_IN_ZERO>(a, p, st);
  } else if (d->KH == 2) {
    launch_wgrad<2, PRN_IN_UP2_PHASE>(a, p, st, 4);
  } else if (d->KH == 3) {
    if (mode == PRN_IN_ZERO) launch_wgrad<3, PRN_IN_ZERO>(a, p, st);
    else if (mode == PRN_IN_REFLECT) launch_wgrad<3, PRN_IN_REFLECT>(a, p, st);
    else launch_wgrad<3, PRN_IN_UP2_REFLECT>(a, p, st);
  } else {
    launch_wgrad<7, PRN_IN_ZERO>(a, p, st);
  }
  PRN_CHECK_LAUNCH("prn_conv2d_wgrad");
  if (phase == 1) return 0;
reduce_only:
  if (p.splits > 1) {
    const int64_t n = (int64_t)g.phases * a.M * a.K;
    for (int r = PRN_REPS(4); r > 0; --r) hipLaunchKernelGGL(reduce_splits_kernel, dim3(cdiv(n, 64)), dim3(256), 0, st, (const float*)ws, dw, n, p.splits);
    PRN_CHECK_LAUNCH("prn_conv2d_wgrad/reduce");
  }
  return 0;
}
}  // namespace

extern "C" int prn_weight_flip_transpose(const float* w, float* wt, int M, int C, int KH, int KW, void* stream) {
  PRN_REQUIRE(w && wt && M > 0 && C > 0 && KH > 0 && KW > 0, "prn_weight_flip_transpose: bad arguments");
  const int64_t n = (int64_t)M * C * KH * KW;
  hipLaunchKernelGGL(flip_transpose_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w, wt, M, C, KH, KW);
  PRN_CHECK_LAUNCH("prn_weight_flip_transpose");
  return 0;
}

// nb independent GEMMs Y_z[M x P] = U_z[M x C] * V_z[C x P] in one launch of the 1x1 (VEC) instances: the product step of
// the Winograd path (prn_winograd.hip), z = one of the 36 transform-domain positions.
extern "C" int64_t prn_gemm_batched_ws_bytes(int M, int C, int P, int nb, const prn_gemm_opts* opts) {
  if (M <= 0 || C <= 0 || P <= 0 || nb <= 0) return -1;
  return prn_split_gemm_plan(M, C, 1, P, nb, opts) == 1 ? prn_split_gemm_image_bytes(M, C, nb) : 0;
}
extern "C" int prn_gemm_batched(int M, int C, int P, int nb, const float* U, const void* u_images, const float* V, float* Y, void* ws, const prn_gemm_opts* opts,
                                void* stream) {
  return prn_gemm_batched_epi(M, C, P, nb, U, u_images, V, Y, ws, opts, PRN_EPI_NONE, stream);
}

// (internal, prn_common.h) the same with an activation in the epilogue: the per-image dynamic convolutions of the plane prior
int prn_gemm_batched_epi(int M, int C, int P, int nb, const float* U, const void* u_images, const float* V, float* Y, void* ws, const prn_gemm_opts* opts, int epi,
                         void* stream) {
  PRN_REQUIRE(U && V && Y && M > 0 && C > 0 && P > 0 && nb > 0 && nb < 65536, "prn_gemm_batched: bad arguments");
  PRN_REQUIRE((P & 3) == 0 && (reinterpret_cast<uintptr_t>(V) & 15) == 0, "prn_gemm_batched: P %% 4 == 0 and 16-byte aligned V required (P=%d)", P);
  PRN_REQUIRE((int64_t)C * P < (1LL << 29) && (int64_t)M * P < (1LL << 29), "prn_gemm_batched: operand larger than a buffer descriptor");
  if ((u_images != nullptr || ws != nullptr) && prn_split_gemm_plan(M, C, 1, P, nb, opts) == 1)      // (neither images nor a workspace to cut into: the fp32 kernel)
    return prn_split_gemm(U, u_images, V, nullptr, nullptr, Y, ws, nullptr, M, C, 1, P, nb, (int64_t)M * C, (int64_t)C * P, (int64_t)M * P, epi, 1, opts,
                          (hipStream_t)stream, 0);
  ConvArgs a;
  a.x = V; a.w = U; a.bias = nullptr; a.addend = nullptr; a.y = Y; a.ws = nullptr;
  a.B = 1; a.C = C; a.H = 1; a.W = P; a.M = M; a.stride = 1; a.pad = 0; a.Ho = 1; a.Wo = P; a.epi = epi;
  a.K = C; a.N = P; a.HoWo = P; a.HW = P; a.xbytes = C * P * 4; a.wbytes = M * C * 4;
  a.ystride = 1; a.yW = P; a.yHW = P;
  a.zx = C * P; a.zw = M * C; a.zy = M * P;
  a.seg.nseg = 0;
  static const prn_env4 forced_ = prn_env_ints("PRN_WINO_TILE");      // PRN_WINO_TILE="tm,tn" (tuning)
  const int* forced = forced_.v;
  FwdPlan p;
  p.wm = 2; p.wn = 2; p.bk = 16; p.splits = 1;
  const int64_t t22 = (int64_t)cdiv(M, 128) * cdiv(P, 128) * nb;
  if (forced[0] > 0) { p.tm = forced[0]; p.tn = forced[1]; }
  else if (M >= 128 && t22 >= 4096) { p.tm = 2; p.tn = 2; }      // tools/winograd_bench.py: 64 x 64 tiles win below ~4096 tiles of 128 x 128
  else { p.tm = 1; p.tn = 1; }
  launch_fwd<1, PRN_IN_ZERO>(a, p, (hipStream_t)stream, nb);
  PRN_CHECK_LAUNCH("prn_gemm_batched");
  return 0;
}

// nb independent products out_z[M x C] = A_z[M x P] * B_z[C x P]^T (reduction over P) on the 1x1 weight-gradient instances:
// the product step of the Winograd weight gradient.  Partials go to ws as [splits][nb][M][C]; the caller sums them
// (prn_winograd.hip folds that sum into the G^T . G transform).  Returns the split count through *splits.
namespace {
// 64 x 64 tiles when the 128 x 128 plan leaves most of the GPU empty (short reductions cap the split count): 54 -> 39 us for
// 36 x [256 x 256 x 640] (tools/winograd_bench.py)
WgPlan plan_batched_nt(const prn_gemm_opts* opts, int M, int C, int P, int nb) {
  const prn_gemm_opts o = prn_opts_or_zero(opts);
  WgPlan p = plan_wgrad(o, M, C, P, nb);
  if (p.tm == 2 && p.tj == 2 && (int64_t)p.tilesM * p.tilesJ * nb * p.splits < 400) p = plan_wgrad(o, M, C, P, nb, false, true);
  return p;
}
}  // namespace
extern "C" int prn_gemm_batched_nt_kind(int M, int C, int P, int nb, const prn_gemm_opts* opts) {      // 0: fp32 MFMA, 2: fp16-piece kernel
  return prn_wgrad16_plan(M, C, P, P, nb, opts) ? 2 : 0;
}
extern "C" int prn_gemm_batched_nt_splits(int M, int C, int P, int nb, const prn_gemm_opts* opts) {
  if (const int s16 = prn_wgrad16_plan(M, C, P, P, nb, opts)) return s16;
  return plan_batched_nt(opts, M, C, P, nb).splits;
}
extern "C" int prn_gemm_batched_nt(int M, int C, int P, int nb, const float* A, const float* Bm, float* ws, const prn_gemm_opts* opts, void* stream) {
  PRN_REQUIRE(A && Bm && ws && M > 0 && C > 0 && P > 0 && nb > 0 && nb < 65536, "prn_gemm_batched_nt: bad arguments");
  PRN_REQUIRE((int64_t)C * P < (1LL << 29) && (int64_t)M * P < (1LL << 29), "prn_gemm_batched_nt: operand larger than a buffer descriptor");
  if (const int s16 = prn_wgrad16_plan(M, C, P, P, nb, opts)) {   // (partials [splits][nb][M][C], the layout of the fp32 kernel)
    // prn_gemm_batched_nt_splits told the caller s16 partial slabs: the plan does not depend on the pointers, a misaligned operand is an error
    PRN_REQUIRE(((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(Bm)) & 15) == 0,
                "prn_gemm_batched_nt: A and B must be 16-byte aligned (the options plan the 16-bit-pipe kernel; opts->wgrad_split = PRN_SPLIT_OFF selects the fp32 kernel)");
    return prn_wgrad16_launch(A, Bm, nullptr, nullptr, 0, ws, M, C, 1, P, nb, (int64_t)M * P, (int64_t)C * P, s16, opts, (hipStream_t)stream);
  }
  WgArgs a;
  a.x = Bm; a.dy = A; a.out = ws;
  a.B = 1; a.C = C; a.H = 1; a.W = P; a.M = M; a.stride = 1; a.pad = 0; a.Ho = 1; a.Wo = P;
  a.K = C; a.N = P; a.HoWo = P; a.HW = P; a.xbytes = C * P * 4; a.dybytes = M * P * 4; a.xz = C * P;
  a.ngroup = 0;
  a.seg.nseg = 0;
  const WgPlan p = plan_batched_nt(opts, M, C, P, nb);
  a.tilesM = p.tilesM; a.tilesJ = p.tilesJ; a.splits = p.splits; a.chunks = p.chunks;
  launch_wgrad<1, PRN_IN_ZERO>(a, p, (hipStream_t)stream, nb);
  PRN_CHECK_LAUNCH("prn_gemm_batched_nt");
  return 0;
}

extern "C" int prn_weight_flip_transpose_batched(const prn_flip_item* items_dev, int n_items, int64_t total_blocks, void* stream) {
  PRN_REQUIRE(items_dev && n_items > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "prn_weight_flip_transpose_batched: bad arguments");
  hipLaunchKernelGGL(flip_transpose_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, n_items, total_blocks);
  PRN_CHECK_LAUNCH("prn_weight_flip_transpose_batched");
  return 0;
}

extern "C" int prn_pad_fold(const float* dp, float* dx, int B, int C, int H, int W, int up2, void* stream) {
  PRN_REQUIRE(dp && dx && B > 0 && C > 0 && H > 1 && W > 1, "prn_pad_fold: bad arguments");
  const int64_t n = (int64_t)B * C * H * W;
  if ((W & 3) == 0 && W >= 8 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0)
    hipLaunchKernelGGL(pad_fold4_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, dp, dx, B * C, H, W, up2, 0);
  else
    hipLaunchKernelGGL(pad_fold_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dp, dx, B * C, H, W, up2, 0);
  PRN_CHECK_LAUNCH("prn_pad_fold");
  return 0;
}

extern "C" int prn_pad_fold_pitched(const float* dp, float* dx, int B, int C, int H, int W, int pitch, void* stream) {
  PRN_REQUIRE(dp && dx && B > 0 && C > 0 && H > 1 && W > 1 && pitch >= W + 2, "prn_pad_fold_pitched: bad arguments");
  const int64_t n = (int64_t)B * C * H * W;
  if ((W & 3) == 0 && W >= 8 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0)
    hipLaunchKernelGGL(pad_fold4_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, dp, dx, B * C, H, W, 0, pitch);
  else
    hipLaunchKernelGGL(pad_fold_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dp, dx, B * C, H, W, 0, pitch);
  PRN_CHECK_LAUNCH("prn_pad_fold_pitched");
  return 0;
}

extern "C" int prn_channel_sum(const float* x, float* out, double* ws, int B, int C, int HW, void* stream) {
  PRN_REQUIRE(x && out && ws && B > 0 && C > 0 && HW > 0, "prn_channel_sum: bad arguments");
  int S = (int)(((int64_t)B * HW + 8191) / 8192);
  S = S > PRN_BN_SPLITS ? PRN_BN_SPLITS : (S < 1 ? 1 : S);
  hipStream_t st = (hipStream_t)stream;
  if (C >= 128 && (int64_t)B * HW <= 65536) S = 1;       // enough channels to fill the GPU on their own: no split, no second launch
  for (int r = PRN_REPS(8); r > 0; --r) {
    hipLaunchKernelGGL(channel_sum_partial_kernel, dim3(C, S), dim3(256), 0, st, x, ws, B, C, HW, S == 1 ? out : nullptr);
    if (S > 1) hipLaunchKernelGGL(channel_sum_final_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, (const double*)ws, out, C, S);
  }
  PRN_CHECK_LAUNCH("prn_channel_sum");
  return 0;
}

// Weight gradient of the plain-GEMM layers on the 16-bit matrix pipe: dW[z][m][c] = sum_n dy[z][m][n] * x[z][c][n]  (reference: the weight
// gradient of every stride-1 1x1 nn.Conv2d, models/backbone.py:56-66, models/fpn.py:46-57, planerecnet.py:510-584; and the 36 transform-domain
// products of the Winograd weight gradient, prn_winograd.hip).  Both operands are ACTIVATIONS, so -- unlike the forward kernel of
// prn_gemm_split.hip, whose weight side is cut once per step -- both have to be cut inside the launch.  The reduction index n (pixels) is the
// unit-stride axis of both operands, which is exactly the MFMA operand layout (a lane of v_mfma_f32_32x32x16_f16 holds eight consecutive k of
// one row), so no transposition is needed anywhere:
//
//   global --buffer_load_dwordx4--> registers: a workgroup's 256 threads fetch the 16-pixel chunk of its 128 dy rows and 128 x rows (four lanes
//            per row, 64 contiguous bytes), one chunk ahead of their use;
//   cut ONCE: every element is scaled by an exact power of two (per ROW, see below), h = fp16(x'), l = fp16(x' - h) (round to nearest; 22 of
//            fp32's 24 significand bits) and the two pieces go to LDS as packed fp16 -- [plane][row][16 px], 32 bytes per row, the two 16-byte
//            halves of a row swapped on every second group of four rows so that the fragment reads below hit 32 different banks;
//   MFMA:    four waves, 64 x 64 of the 128 x 128 output tile each; a wave reads its fragments with eight ds_read_b128 per chunk (each IS an
//            MFMA operand: 8 consecutive pixels of one row) and issues 2 x 2 x 3 products l*h, h*l, h*h with fp32 accumulation.
//   One barrier per chunk; pieces double-buffered.
//
// Scaling.  A row's elements are multiplied by 2^(11 - E), E = the RUNNING frexp exponent of the row's largest element so far, raised only
// when a chunk's largest element exceeds it by more than three binades (then E := that chunk's exponent), so |x'| < 2^14 always (fp16's range
// ends at 2^16).  When E of a dy row / an x row rises, the accumulators of that output row / column are brought to the new scale by an exact
// v_ldexp_f32 before the next product (rare after the first chunks; a wave-uniform branch).  The epilogue undoes both scalings:
// dW[m][c] = acc * 2^(E_dy[m] + E_x[c] - 22).  Elements more than 2^14 below their row's running maximum lose RELATIVE precision (their second
// piece leaves fp16's normal range) at an absolute error of 2^-36 of that maximum: the forward kernel's property, per row instead of per
// column (tests/test_ops_gpu.py::test_wgrad16_*: error against fp64 next to the fp32 kernel's on the same operands).
//
// The pixel axis is split over workgroups (blockIdx.y); partial tiles go to the caller's workspace [splits][nz][M][C] and are summed in a
// fixed order by the existing reduce kernels (prn_conv.hip) / by winograd_dw_kernel.
#include "prn_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
typedef __attribute__((ext_vector_type(2))) float f2_t;
typedef __attribute__((ext_vector_type(16))) float f16x_t;

constexpr unsigned W16_OOB = 0x80000000u;
constexpr int W16_PLANE = 128 * 32;              // bytes of one piece plane: 128 rows x 16 px x 2 B
constexpr int W16_BUF = 4 * W16_PLANE;           // dy.h, dy.l, x.h, x.l

struct Wg16Args {
  const float* dy; const float* x; float* out;
  int M, C, HW, B;                 // rows of dy / of x per image, pixels per image, images
  int N;                           // B * HW
  int tilesM, tilesC, splits, chunks;
  long long zdy, zx;               // element strides per blockIdx.z (batched products); 0 for a convolution
  long long zout;                  // M * C
  int nz;
  int wide;                        // epilogue through the LDS transpose: C % 4 == 0 and a 16-byte aligned output
  int ngroup;                      // > 0: blockIdx.z = layer of a group of same-shape layers, operands from the tables below
  const float* gx[PRN_WGRAD_GROUP_MAX];
  const float* gdy[PRN_WGRAD_GROUP_MAX];
};

__device__ __forceinline__ const float* w16_kernarg_ptr(size_t member_offset, int i) {
  typedef __attribute__((address_space(4))) const uint64_t* kptr;
  kptr base = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
  return reinterpret_cast<const float*>(base[member_offset / 8 + i]);
}

// max over the four lanes of a quad (the four 4-pixel quarters of one row's chunk)
__device__ __forceinline__ float quad_max(float v) {
  const int a = __builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
  v = fmaxf(v, __int_as_float(a));
  const int b = __builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
  return fmaxf(v, __int_as_float(b));
}

// NP = 3: l*h, h*l, h*h;  4: + l*l
template <int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void wgrad16_kernel(const Wg16Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char pieces[2 * W16_BUF];
  __shared__ int rowexp[2][256];                 // per chunk buffer: the exponent E each of the 256 rows' pieces were scaled with
  __shared__ int scratch[4][64];                 // per wave: row deltas / final exponents for the accumulator rescale
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // One-dimensional grid, renumbered so that the tilesM x tilesC output tiles of one (pixel split, z) -- which all stream the SAME dy / x pixel
  // range -- are neighbours on ONE XCD (block b runs on XCD b % 8; each XCD has its own L2): the operand chunk is then fetched from HBM once
  // per XCD and served to the other tiles from its L2.  Spread over the XCDs (the old (tile, split, z) grid) every tile fetched its own copy:
  // 157 MB of reads per stage-3 layer for 49 MB of operands.
  const int tiles = a.tilesM * a.tilesC;
  const int lid = prn_xcd_remap(blockIdx.x, tiles * a.splits * a.nz);
  const int tile = lid % tiles, rest = lid / tiles;
  const int tm = tile % a.tilesM, tc = tile / a.tilesM;
  const int sp = rest % a.splits, z = rest / a.splits;
  const int q0 = (int)((long long)a.chunks * sp / a.splits), q1 = (int)((long long)a.chunks * (sp + 1) / a.splits);
  const int HW = a.HW;
  const float* dyz = a.ngroup > 0 ? w16_kernarg_ptr(offsetof(Wg16Args, gdy), z) : a.dy + (long long)z * a.zdy;
  const float* xz = a.ngroup > 0 ? w16_kernarg_ptr(offsetof(Wg16Args, gx), z) : a.x + (long long)z * a.zx;
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dyz), 0, a.B * a.M * HW * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xz), 0, a.B * a.C * HW * 4, 0x00020000);

  // ---- producer side: thread t owns the 4-pixel quarter (t & 3) of rows (t >> 2) + 64 k, k = 0, 1 (dy rows) and k = 2, 3 (x rows)
  const int quarter = t & 3, rsub = t >> 2;
  unsigned rowoff[4];                            // byte offset of the row inside an image, or the out-of-range marker
  int erun[4];                                   // running exponent of the row (identical in the four lanes of its quad)
  unsigned wroff[4];                             // LDS byte offset of this quarter inside a piece plane pair
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int rl = rsub + 64 * (k & 1);
    const int row = (k < 2 ? tm : tc) * 128 + rl;
    const bool ok = row < (k < 2 ? a.M : a.C);
    rowoff[k] = ok ? (unsigned)(row * HW * 4) : W16_OOB;
    erun[k] = -200;
    wroff[k] = (unsigned)((k < 2 ? 0 : 2 * W16_PLANE) + rl * 32 + (((quarter >> 1) ^ ((rl >> 2) & 1)) * 16) + (quarter & 1) * 8);
  }
  const unsigned imgA = (unsigned)(a.M * HW * 4), imgB = (unsigned)(a.C * HW * 4);
  // pixel cursor of this thread's quarter: global pixel n = q * 16 + quarter * 4 -> (image b, pixel p); HW % 4 == 0 keeps a quarter inside one image
  int cb, cp;
  {
    const long long n = (long long)q0 * 16 + quarter * 4;
    cb = (int)(n / HW); cp = (int)(n - (long long)cb * HW);
  }
  // Two chunks of loads in flight per thread (register sets ld0 / ld1, chunk c lives in set c & 1): with one, the launch ran at memory LATENCY --
  // 3 workgroups x 16 KB in flight per CU is 12.6 MB over the GPU, i.e. 6.3 TB/s at ~2 us under load, which is what it measured.
  float4 ld0[4], ld1[4];
  auto load_chunk = [&](float4 (&ld)[4]) {       // the chunk at the cursor -> ld[], cursor += 16 pixels
    // branch-free: an invalid row / a pixel past the tensor gets bit 31 set in its offset (the descriptor's range check then returns zeros
    // without a memory access); rowoff already carries that bit for rows outside the tile
    const unsigned past = (unsigned)((a.B - 1 - cb) >> 31) << 31;          // cb >= B  ->  0x80000000
    const unsigned pa = ((unsigned)cb * imgA + (unsigned)cp * 4u) | past, pb = ((unsigned)cb * imgB + (unsigned)cp * 4u) | past;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      ld[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(k < 2 ? ars : brs, (int)((k < 2 ? pa : pb) + (rowoff[k] & 0x7fffffffu)) | (int)(rowoff[k] & 0x80000000u), 0, 0));
    cp += 16;
    const int wrap = (HW - 1 - cp) >> 31;                                  // cp >= HW -> -1   (HW >= 16: at most one image boundary per chunk)
    cp -= HW & wrap; cb -= wrap;
  };
  auto cut_chunk = [&](const float4 (&ld)[4], int buf) {      // ld[] -> pieces[buf], rowexp[buf]
    unsigned char* pb = pieces + buf * W16_BUF;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = ld[k];
      const float mx = quad_max(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      int ec = mx > 0.f ? __builtin_amdgcn_frexp_expf(mx) : -200;
      ec = ec < -100 ? (mx > 0.f ? -100 : -200) : ec;                 // (keeps 2^(11 - E) a normal float; |x| < 2^-100 cuts to zero pieces)
      if (ec > erun[k] + 3) erun[k] = ec;
      const float s = __builtin_amdgcn_ldexpf(1.0f, 11 - (erun[k] < -100 ? -100 : erun[k]));
      const f2_t s2 = {s, s};
      const f2_t x0 = f2_t{v.x, v.y} * s2, x1 = f2_t{v.z, v.w} * s2;
      const h2_t h0 = __builtin_convertvector(x0, h2_t), h1 = __builtin_convertvector(x1, h2_t);
      const f2_t r0 = x0 - __builtin_convertvector(h0, f2_t), r1 = x1 - __builtin_convertvector(h1, f2_t);
      const h2_t l0 = __builtin_convertvector(r0, h2_t), l1 = __builtin_convertvector(r1, h2_t);
      uint2 hh, ll;
      hh.x = __builtin_bit_cast(unsigned, h0); hh.y = __builtin_bit_cast(unsigned, h1);
      ll.x = __builtin_bit_cast(unsigned, l0); ll.y = __builtin_bit_cast(unsigned, l1);
      *reinterpret_cast<uint2*>(pb + wroff[k]) = hh;
      *reinterpret_cast<uint2*>(pb + wroff[k] + W16_PLANE) = ll;
      if (quarter == 0) rowexp[buf][rsub + 64 * k] = erun[k];
    }
  };

  // ---- consumer side: wave (wm, wn) owns output rows wm * 64 .. + 63 (dy rows) x columns wn * 64 .. + 63 (x rows)
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, g = lane >> 5;
  f16x_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  int prevA = -200, prevB = -200;                // exponent the accumulators' row (wm * 64 + lane) / column (wn * 64 + lane) currently carry
  auto mma_chunk = [&](int buf) {
    const int newA = rowexp[buf][wm * 64 + lane], newB = rowexp[buf][128 + wn * 64 + lane];
    const int dA = prevA - newA, dB = prevB - newB;                   // <= 0
    prevA = newA; prevB = newB;
    if (__builtin_amdgcn_ballot_w64((dA | dB) != 0) != 0ull) {        // a row's maximum grew: bring its partial sums to the new scale (exact)
      scratch[wave][lane] = dA;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int db = __shfl(dB, j * 32 + r, 64);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int da = scratch[wave][i * 32 + (e >> 2) * 8 + g * 4 + (e & 3)];
            acc[i][j][e] = __builtin_amdgcn_ldexpf(acc[i][j][e], da + db);
          }
      }
      __builtin_amdgcn_wave_barrier();
    }
    const unsigned char* pb = pieces + buf * W16_BUF;
    h8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ra = wm * 64 + u * 32 + r, rb = wn * 64 + u * 32 + r;
      const unsigned oa = (unsigned)(ra * 32 + ((g ^ ((ra >> 2) & 1)) * 16)), ob = (unsigned)(2 * W16_PLANE + rb * 32 + ((g ^ ((rb >> 2) & 1)) * 16));
      ah[u] = *reinterpret_cast<const h8_t*>(pb + oa); al[u] = *reinterpret_cast<const h8_t*>(pb + oa + W16_PLANE);
      bh[u] = *reinterpret_cast<const h8_t*>(pb + ob); bl[u] = *reinterpret_cast<const h8_t*>(pb + ob + W16_PLANE);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f16x_t c = acc[i][j];
        if (NP >= 4) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bl[j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], c, 0, 0, 0);
        acc[i][j] = c;
      }
  };

  const int nq = q1 - q0;
  if (nq > 0) {
    // (loads past the last chunk are issued unconditionally: their pixel cursor is beyond the tensor, the descriptor returns zeros without a
    // memory access, and the loop body stays branch-free)
    load_chunk(ld0);                             // chunk 0
    load_chunk(ld1);                             // chunk 1
    cut_chunk(ld0, 0);
    load_chunk(ld0);                             // chunk 2
    __syncthreads();
    int i = 0;
    for (; i + 1 < nq; i += 2) {
      cut_chunk(ld1, 1);                         // chunk i + 1: registers -> LDS (its loads were issued two iterations ago)
      load_chunk(ld1);                           // chunk i + 3
      __builtin_amdgcn_sched_barrier(0);
      mma_chunk(0);                              // chunk i
      __syncthreads();
      cut_chunk(ld0, 0);                         // chunk i + 2
      load_chunk(ld0);                           // chunk i + 4
      __builtin_amdgcn_sched_barrier(0);
      mma_chunk(1);                              // chunk i + 1
      __syncthreads();
    }
    if (i < nq) mma_chunk(0);                    // odd count: the last chunk was cut by the previous iteration (or the prologue)
  }

  // ---- epilogue: undo both scalings; partial tile (splits > 1) or the result itself
  scratch[wave][lane] = prevA;
  __builtin_amdgcn_wave_barrier();
  float* ob = a.out + ((long long)sp * a.nz + z) * a.zout;       // partial [split][z][M][C]
  if (a.wide) {
    // through an LDS transpose (the piece buffers are dead): a lane then owns four consecutive columns of one row -- dwordx4 stores instead
    // of 64 dword stores per lane (the same epilogue that took 2.6 % off the training step in split16_gemm_kernel)
    __syncthreads();
    float* cw = reinterpret_cast<float*>(pieces) + wave * (32 * 36);
    const int crow = lane >> 3, ccol = (lane & 7) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int eb = __shfl(prevB, j * 32 + r, 64);
      const int col4 = tc * 128 + wn * 64 + j * 32 + ccol;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ro = (e >> 2) * 8 + g * 4 + (e & 3);
          const int ea = scratch[wave][i * 32 + ro];
          cw[ro * 36 + r] = (ea <= -200 || eb <= -200) ? 0.f : __builtin_amdgcn_ldexpf(acc[i][j][e], ea + eb - 22);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rl = crow + 8 * q, row = tm * 128 + wm * 64 + i * 32 + rl;
          const float4 v = *reinterpret_cast<const float4*>(&cw[rl * 36 + ccol]);
          if (row < a.M && col4 < a.C) *reinterpret_cast<float4*>(ob + (long long)row * a.C + col4) = v;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = tc * 128 + wn * 64 + j * 32 + r;
    const int eb = __shfl(prevB, j * 32 + r, 64);
    if (col >= a.C) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rl = i * 32 + (e >> 2) * 8 + g * 4 + (e & 3), row = tm * 128 + wm * 64 + rl;
        const int ea = scratch[wave][rl];
        if (row < a.M) ob[(long long)row * a.C + col] = (ea <= -200 || eb <= -200) ? 0.f : __builtin_amdgcn_ldexpf(acc[i][j][e], ea + eb - 22);
      }
  }
}

}  // namespace

// ---- internal interface (prn_common.h) -----------------------------------------------------------------------------------------------
// Plan: number of pixel splits for dW[nz][M][C] over N pixels on the 16-bit kernel, 0 = keep the fp32 kernel.
int prn_wgrad16_plan(int M, int C, int64_t N, int HW, int nz, const prn_gemm_opts* o) {
  if (o == nullptr || o->wgrad_split == PRN_SPLIT_OFF) return 0;
  if (M <= 0 || C <= 0 || N <= 0 || nz <= 0 || nz >= 65536 || (HW & 3) != 0 || HW < 16 || N % HW != 0) return 0;
  if ((int64_t)M * N >= (1LL << 29) || (int64_t)C * N >= (1LL << 29)) return 0;
  const int64_t tiles = (int64_t)cdiv(M, 128) * cdiv(C, 128) * nz;
  const int chunks = cdiv(N, 16);
  if (o->wgrad_split != PRN_SPLIT_ALWAYS) {
    // where it pays (tools/wgrad16_bench.py): both tile dimensions mostly full and enough work to amortise the partial-sum round trip
    if ((int64_t)cdiv(M, 128) * 128 * 4 > (int64_t)M * 5 || (int64_t)cdiv(C, 128) * 128 * 4 > (int64_t)C * 5) return 0;      // > 1/5 of a tile dimension padding
    if (2.0 * M * C * (double)N * nz < 0.25e9 * (double)o->split_min_gflop) return 0;      // (1 GFLOP at the default floor of 4)
  }
  const int target = o->wgrad_wgs > 0 ? o->wgrad_wgs : 768;            // three workgroups per CU (168 VGPRs): one full residency round
  int64_t s = tiles >= target ? 1 : target / tiles;
  const int smax = chunks / 8 > 0 ? chunks / 8 : 1;                 // at least 128 pixels per split
  if (s > smax) s = smax;
  if (s > 256) s = 256;
  s = prn_quantise_splits(tiles, (int)s);
  return s < 1 ? 1 : (int)s;
}

// out: dw when splits == 1, else the partial sums [splits][nz][M][C] (the caller sums them).  x / dy: dense [B][C|M][HW] per z; gx / gdy
// (ngroup > 0): host arrays of ngroup device pointers, blockIdx.z = layer.
int prn_wgrad16_launch(const float* dy, const float* x, const float* const* gdy, const float* const* gx, int ngroup, float* out, int M, int C, int B, int HW, int nz,
                       int64_t zdy, int64_t zx, int splits, const prn_gemm_opts* o, hipStream_t st) {
  PRN_REQUIRE(out && splits >= 1 && (ngroup > 0 ? (gdy && gx && ngroup <= PRN_WGRAD_GROUP_MAX) : (dy && x)), "prn_wgrad16: bad arguments");
  Wg16Args a;
  a.dy = dy; a.x = x; a.out = out; a.M = M; a.C = C; a.HW = HW; a.B = B; a.N = B * HW;
  a.tilesM = cdiv(M, 128); a.tilesC = cdiv(C, 128); a.splits = splits; a.chunks = cdiv((int64_t)B * HW, 16);
  a.zdy = zdy; a.zx = zx; a.zout = (long long)M * C; a.nz = ngroup > 0 ? ngroup : nz; a.ngroup = ngroup;
  for (int i = 0; i < PRN_WGRAD_GROUP_MAX; ++i) {
    a.gx[i] = ngroup > 0 ? gx[i < ngroup ? i : 0] : nullptr; a.gdy[i] = ngroup > 0 ? gdy[i < ngroup ? i : 0] : nullptr;
    if (ngroup > 0) {
      PRN_REQUIRE(a.gx[i] && a.gdy[i] && (reinterpret_cast<uintptr_t>(a.gx[i]) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.gdy[i]) & 15) == 0, "prn_wgrad16: null / unaligned tensor in the group");
    }
  }
  a.wide = (C & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  const dim3 grid((unsigned)(a.tilesM * a.tilesC * splits * a.nz)), block(256);
  if (o && o->split_products >= 4) hipLaunchKernelGGL(wgrad16_kernel<4>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(wgrad16_kernel<3>, grid, block, 0, st, a);
  PRN_CHECK_LAUNCH("prn_wgrad16");
  return 0;
}

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
#include "prn_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct ConvArgs {
  const float* x; const float* w; const float* bias; const float* addend; float* y;
  int B, C, H, W, M, stride, pad, Ho, Wo, mode, epi;
  int K, N, HoWo, HW, tilesM, nblocks;
};

__device__ __forceinline__ int reflect_idx(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * n - 2 - i : i;
}

// One element of the virtual im2col matrix. (coff = c*H*W, r, s) identify the K row, (ih0, iw0) the pixel.
__device__ __forceinline__ float gather_px(const float* __restrict__ xb, int coff, int r, int s, int ih0, int iw0,
                                           bool ok, int mode, int H, int W) {
  int ih = ih0 + r, iw = iw0 + s;
  if (mode == PRN_IN_ZERO) {
    ok = ok && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
  } else if (mode == PRN_IN_REFLECT) {
    ih = reflect_idx(ih, H);
    iw = reflect_idx(iw, W);
  } else if (mode == PRN_IN_UP2_REFLECT) {
    ih = reflect_idx(ih, 2 * H) >> 1;
    iw = reflect_idx(iw, 2 * W) >> 1;
  } else {  // PRN_IN_DILATED (factor 2): only even virtual coordinates carry data
    ok = ok && ih >= 0 && iw >= 0 && ((ih | iw) & 1) == 0;
    ih >>= 1;
    iw >>= 1;
    ok = ok && ih < H && iw < W;
  }
  float v = 0.f;
  if (ok) v = xb[(size_t)coff + (size_t)(ih * W + iw)];
  return v;
}

template <int KS, int TM, int TN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 16, LDA = 17;
  constexpr int KSTEP = 256 / BN;  // K rows covered by one sweep of the block
  constexpr int NB = BK / KSTEP;   // gathered elements per thread per K slice
  constexpr int KK = KS * KS;
  __shared__ float As[2][BM * LDA];
  __shared__ float Bs[2][BK * BN];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int id = prn_xcd_remap(blockIdx.x, a.nblocks);
  const int m0 = (id % a.tilesM) * BM, n0 = (id / a.tilesM) * BN;

  // pixel owned by this thread in the B (im2col) operand
  const int nl = tid % BN, krow0 = tid / BN;
  const int n = n0 + nl;
  const bool nvalid = n < a.N;
  int b = 0, oh = 0, ow = 0;
  if (nvalid) {
    b = n / a.HoWo;
    const int p = n - b * a.HoWo;
    oh = p / a.Wo;
    ow = p - oh * a.Wo;
  }
  const int ih0 = oh * a.stride - a.pad, iw0 = ow * a.stride - a.pad;
  const float* __restrict__ xb = a.x + (size_t)b * a.C * a.HW;

  const int arow = tid >> 2, akq = (tid & 3) * 4;
  const bool k4 = ((a.K & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.w) & 15) == 0);

  float ra[TM][4];
  float rb[NB];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + arow + 64 * i, k = k0 + akq;
      const float* wp = a.w + (size_t)m * a.K + k;
      if (m < a.M && k4 && k < a.K) {
        const float4 v = *reinterpret_cast<const float4*>(wp);
        ra[i][0] = v.x; ra[i][1] = v.y; ra[i][2] = v.z; ra[i][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[i][j] = (m < a.M && k + j < a.K) ? wp[j] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int k = k0 + krow0 + i * KSTEP;
      const int c = k / KK, rs = k - c * KK, r = rs / KS, s = rs - r * KS;
      rb[i] = gather_px(xb, c * a.HW, r, s, ih0, iw0, nvalid && k < a.K, a.mode, a.H, a.W);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) As[buf][(arow + 64 * i) * LDA + akq + j] = ra[i][j];
#pragma unroll
    for (int i = 0; i < NB; ++i) Bs[buf][(krow0 + i * KSTEP) * BN + nl] = rb[i];
  };

  const int KT = (a.K + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) load_tile((kt + 1) * BK);
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
    if (kt + 1 < KT) store_tile(buf ^ 1);
    __syncthreads();
  }

  // epilogue: C/D layout col = lane&31 (pixel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (channel)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nn = n0 + wn * TN * 32 + j * 32 + (lane & 31);
    if (nn >= a.N) continue;
    const int bb = nn / a.HoWo, p = nn - bb * a.HoWo;
    const size_t base = (size_t)bb * a.M * a.HoWo + p;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < a.M) {
          const size_t idx = base + (size_t)m * a.HoWo;
          float v = acc[i][j][r];
          if (a.bias) v += a.bias[m];
          if (a.addend) v += a.addend[idx];
          if (a.epi == PRN_EPI_RELU) v = fmaxf(v, 0.f);
          else if (a.epi == PRN_EPI_SIGMOID) v = 1.f / (1.f + __expf(-v));
          a.y[idx] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- wgrad
struct WgArgs {
  const float* x; const float* dy; float* out;
  int B, C, H, W, M, stride, pad, Ho, Wo, mode;
  int K, N, HoWo, HW, tilesM, tilesJ, splits, chunks;
};

template <int KS, int TM, int TJ>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgArgs a) {
  constexpr int BM = 64 * TM, BJ = 64 * TJ, LD = 17, KK = KS * KS;
  constexpr int NBJ = BJ / 16;
  __shared__ float As[2][BM * LD];
  __shared__ float Bs[2][BJ * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wj = wave & 1;
  const int m0 = (blockIdx.x % a.tilesM) * BM, j0 = (blockIdx.x / a.tilesM) * BJ;
  const int split = blockIdx.y;
  const int cbeg = (int)((int64_t)split * a.chunks / a.splits), cend = (int)((int64_t)(split + 1) * a.chunks / a.splits);

  const int arow = tid >> 2, anq = (tid & 3) * 4;
  const bool n4 = ((a.HoWo & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.dy) & 15) == 0);
  const int nl = tid & 15, jrow = tid >> 4;
  int jcoff[NBJ], jr[NBJ], js[NBJ];
  bool jok[NBJ];
#pragma unroll
  for (int i = 0; i < NBJ; ++i) {
    const int j = j0 + jrow + 16 * i;
    const int c = j / KK, rs = j - c * KK;
    jok[i] = j < a.K;
    jcoff[i] = c * a.HW;
    jr[i] = rs / KS;
    js[i] = rs - jr[i] * KS;
  }

  float ra[TM][4], rb[NBJ];
  f32x16 acc[TM][TJ];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto load_chunk = [&](int ch) {
    {  // dY rows: 4 consecutive pixels of one output channel
      const int n = ch * 16 + anq;
      if (n4) {
        const bool ok = n < a.N;
        int b = 0, p = 0;
        if (ok) { b = n / a.HoWo; p = n - b * a.HoWo; }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int m = m0 + arow + 64 * i;
          if (ok && m < a.M) {
            const float4 v = *reinterpret_cast<const float4*>(a.dy + ((size_t)b * a.M + m) * a.HoWo + p);
            ra[i][0] = v.x; ra[i][1] = v.y; ra[i][2] = v.z; ra[i][3] = v.w;
          } else {
            ra[i][0] = ra[i][1] = ra[i][2] = ra[i][3] = 0.f;
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int nq = n + q;
          const bool ok = nq < a.N;
          int b = 0, p = 0;
          if (ok) { b = nq / a.HoWo; p = nq - b * a.HoWo; }
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int m = m0 + arow + 64 * i;
            ra[i][q] = (ok && m < a.M) ? a.dy[((size_t)b * a.M + m) * a.HoWo + p] : 0.f;
          }
        }
      }
    }
    {  // im2col rows
      const int n = ch * 16 + nl;
      const bool ok = n < a.N;
      int b = 0, oh = 0, ow = 0;
      if (ok) {
        b = n / a.HoWo;
        const int p = n - b * a.HoWo;
        oh = p / a.Wo;
        ow = p - oh * a.Wo;
      }
      const int ih0 = oh * a.stride - a.pad, iw0 = ow * a.stride - a.pad;
      const float* __restrict__ xb = a.x + (size_t)b * a.C * a.HW;
#pragma unroll
      for (int i = 0; i < NBJ; ++i) rb[i] = gather_px(xb, jcoff[i], jr[i], js[i], ih0, iw0, ok && jok[i], a.mode, a.H, a.W);
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) As[buf][(arow + 64 * i) * LD + anq + q] = ra[i][q];
#pragma unroll
    for (int i = 0; i < NBJ; ++i) Bs[buf][(jrow + 16 * i) * LD + nl] = rb[i];
  };

  if (cbeg < cend) {
    load_chunk(cbeg);
    store_chunk(0);
  }
  __syncthreads();
  for (int ch = cbeg; ch < cend; ++ch) {
    const int buf = (ch - cbeg) & 1;
    if (ch + 1 < cend) load_chunk(ch + 1);
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
    if (ch + 1 < cend) store_chunk(buf ^ 1);
    __syncthreads();
  }

  float* out = a.out + (size_t)split * a.M * a.K;
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

__global__ void reduce_splits_kernel(const float* __restrict__ ws, float* __restrict__ out, int64_t n, int splits) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += ws[(size_t)k * n + i];
  out[i] = s;
}

__global__ void flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int M, int C, int KH, int KW) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // index into wt [C][M][KH][KW]
  const int64_t total = (int64_t)M * C * KH * KW;
  if (i >= total) return;
  const int s = i % KW, r = (i / KW) % KH, m = (i / (KW * KH)) % M, c = i / ((int64_t)KW * KH * M);
  wt[i] = w[(((size_t)m * C + c) * KH + (KH - 1 - r)) * KW + (KW - 1 - s)];
}

// dx[b,c,h,w] = sum over the virtual padded positions that gather from (h,w)
__global__ void pad_fold_kernel(const float* __restrict__ dp, float* __restrict__ dx, int BC, int H, int W, int up2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)BC * H * W) return;
  const int w = i % W, h = (i / W) % H;
  const int64_t bc = i / ((int64_t)W * H);
  const int Hv = up2 ? 2 * H : H, Wv = up2 ? 2 * W : W;     // virtual (pre-pad) size
  const int Wp = Wv + 2;
  const float* p = dp + bc * (int64_t)(Hv + 2) * Wp;
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
  dx[i] = acc;
}

// out[c] = sum_{b,hw} x[b,c,hw] : one block per channel, fp64 block combine
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int C, int HW) {
  const int c = blockIdx.x;
  double s = 0.0;
  for (int b = 0; b < B; ++b) {
    const float* p = x + ((size_t)b * C + c) * HW;
    float part = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) part += p[i];
    s += (double)part;
  }
  s = wave_sum_d(s);
  __shared__ double sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[c] = (float)(sm[0] + sm[1] + sm[2] + sm[3]);
}

template <int KS>
int launch_fwd(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  auto ntiles = [&](int tm, int tn) { return (int64_t)cdiv(a.M, 64 * tm) * cdiv(a.N, 64 * tn); };
  int tm = 1, tn = 1;
  if (a.M > 64 && ntiles(2, 2) >= 384) { tm = 2; tn = 2; }
  else if (ntiles(1, 2) >= 384) { tm = 1; tn = 2; }
  if (KS == 7) { tm = 1; tn = 2; }
  a.tilesM = cdiv(a.M, 64 * tm);
  a.nblocks = a.tilesM * cdiv(a.N, 64 * tn);
  dim3 grid(a.nblocks), block(256);
  if (KS == 7) hipLaunchKernelGGL((conv_igemm_kernel<KS, 1, 2>), grid, block, 0, st, a);
  else if (tm == 2) hipLaunchKernelGGL((conv_igemm_kernel<KS == 7 ? 3 : KS, 2, 2>), grid, block, 0, st, a);
  else if (tn == 2) hipLaunchKernelGGL((conv_igemm_kernel<KS == 7 ? 3 : KS, 1, 2>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((conv_igemm_kernel<KS == 7 ? 3 : KS, 1, 1>), grid, block, 0, st, a);
  return 0;
}

struct WgPlan { int tm, tj, tilesM, tilesJ, splits, chunks; };
WgPlan plan_wgrad(int M, int K, int64_t N) {
  WgPlan p;
  p.tm = (M > 64) ? 2 : 1;
  p.tj = (K > 64) ? 2 : 1;
  p.tilesM = cdiv(M, 64 * p.tm);
  p.tilesJ = cdiv(K, 64 * p.tj);
  p.chunks = cdiv(N, 16);
  const int tiles = p.tilesM * p.tilesJ;
  int s = cdiv(1024, tiles);
  const int smax = p.chunks / 8 > 0 ? p.chunks / 8 : 1;   // at least 128 pixels per split
  if (s > smax) s = smax;
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  p.splits = s;
  return p;
}

template <int KS>
int launch_wgrad(WgArgs a, const WgPlan& p, hipStream_t st) {
  dim3 grid(p.tilesM * p.tilesJ, p.splits), block(256);
  if (p.tm == 2 && p.tj == 2) hipLaunchKernelGGL((conv_wgrad_kernel<KS, 2, 2>), grid, block, 0, st, a);
  else if (p.tm == 2) hipLaunchKernelGGL((conv_wgrad_kernel<KS, 2, 1>), grid, block, 0, st, a);
  else if (p.tj == 2) hipLaunchKernelGGL((conv_wgrad_kernel<KS, 1, 2>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((conv_wgrad_kernel<KS, 1, 1>), grid, block, 0, st, a);
  return 0;
}

int check_desc(const prn_conv_desc* d, const char* who) {
  PRN_REQUIRE(d != nullptr, "%s: null descriptor", who);
  PRN_REQUIRE(d->KH == d->KW && (d->KH == 1 || d->KH == 3 || d->KH == 7), "%s: kernel %dx%d unsupported (1,3,7 square)", who, d->KH, d->KW);
  PRN_REQUIRE(d->B > 0 && d->C > 0 && d->H > 0 && d->W > 0 && d->M > 0 && d->Ho > 0 && d->Wo > 0, "%s: empty dimension", who);
  PRN_REQUIRE(d->in_mode >= 0 && d->in_mode <= 3, "%s: bad in_mode %d", who, d->in_mode);
  PRN_REQUIRE(d->in_mode != PRN_IN_DILATED || d->dil == 2, "%s: only dilation 2 is implemented", who);
  PRN_REQUIRE((d->in_mode != PRN_IN_REFLECT && d->in_mode != PRN_IN_UP2_REFLECT) || (d->pad == 1 && d->stride == 1 && d->KH == 3),
              "%s: reflect modes need a 3x3 stride-1 pad-1 conv", who);
  PRN_REQUIRE((int64_t)d->B * d->Ho * d->Wo < (1LL << 31) && (int64_t)d->C * d->H * d->W < (1LL << 31), "%s: tensor too large for int32 pixel index", who);
  return 0;
}

}  // namespace

extern "C" int prn_conv2d_fwd(const prn_conv_desc* d, const float* x, const float* w, const float* bias,
                              const float* addend, float* y, void* stream) {
  if (int e = check_desc(d, "prn_conv2d_fwd")) return e;
  PRN_REQUIRE(x && w && y, "prn_conv2d_fwd: null tensor");
  ConvArgs a;
  a.x = x; a.w = w; a.bias = bias; a.addend = addend; a.y = y;
  a.B = d->B; a.C = d->C; a.H = d->H; a.W = d->W; a.M = d->M; a.stride = d->stride; a.pad = d->pad;
  a.Ho = d->Ho; a.Wo = d->Wo; a.mode = d->in_mode; a.epi = d->epilogue;
  a.K = d->C * d->KH * d->KW; a.N = d->B * d->Ho * d->Wo; a.HoWo = d->Ho * d->Wo; a.HW = d->H * d->W;
  hipStream_t st = (hipStream_t)stream;
  if (d->KH == 1) launch_fwd<1>(a, st);
  else if (d->KH == 3) launch_fwd<3>(a, st);
  else launch_fwd<7>(a, st);
  PRN_CHECK_LAUNCH("prn_conv2d_fwd");
  return 0;
}

extern "C" int64_t prn_conv2d_wgrad_ws_bytes(const prn_conv_desc* d) {
  if (check_desc(d, "prn_conv2d_wgrad_ws_bytes")) return -1;
  const int K = d->C * d->KH * d->KW;
  WgPlan p = plan_wgrad(d->M, K, (int64_t)d->B * d->Ho * d->Wo);
  return p.splits > 1 ? (int64_t)p.splits * d->M * K * 4 : 0;
}

extern "C" int prn_conv2d_wgrad(const prn_conv_desc* d, const float* x, const float* dy, float* dw, void* ws, void* stream) {
  if (int e = check_desc(d, "prn_conv2d_wgrad")) return e;
  PRN_REQUIRE(d->in_mode != PRN_IN_DILATED, "prn_conv2d_wgrad: dilated input mode is a dgrad-only mode");
  PRN_REQUIRE(x && dy && dw, "prn_conv2d_wgrad: null tensor");
  WgArgs a;
  a.x = x; a.dy = dy;
  a.B = d->B; a.C = d->C; a.H = d->H; a.W = d->W; a.M = d->M; a.stride = d->stride; a.pad = d->pad;
  a.Ho = d->Ho; a.Wo = d->Wo; a.mode = d->in_mode;
  a.K = d->C * d->KH * d->KW; a.N = d->B * d->Ho * d->Wo; a.HoWo = d->Ho * d->Wo; a.HW = d->H * d->W;
  WgPlan p = plan_wgrad(a.M, a.K, a.N);
  a.tilesM = p.tilesM; a.tilesJ = p.tilesJ; a.splits = p.splits; a.chunks = p.chunks;
  PRN_REQUIRE(p.splits == 1 || ws != nullptr, "prn_conv2d_wgrad: workspace required (%d splits)", p.splits);
  a.out = p.splits > 1 ? (float*)ws : dw;
  hipStream_t st = (hipStream_t)stream;
  if (d->KH == 1) launch_wgrad<1>(a, p, st);
  else if (d->KH == 3) launch_wgrad<3>(a, p, st);
  else launch_wgrad<7>(a, p, st);
  PRN_CHECK_LAUNCH("prn_conv2d_wgrad");
  if (p.splits > 1) {
    const int64_t n = (int64_t)a.M * a.K;
    hipLaunchKernelGGL(reduce_splits_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, (const float*)ws, dw, n, p.splits);
    PRN_CHECK_LAUNCH("prn_conv2d_wgrad/reduce");
  }
  return 0;
}

extern "C" int prn_weight_flip_transpose(const float* w, float* wt, int M, int C, int KH, int KW, void* stream) {
  PRN_REQUIRE(w && wt && M > 0 && C > 0 && KH > 0 && KW > 0, "prn_weight_flip_transpose: bad arguments");
  const int64_t n = (int64_t)M * C * KH * KW;
  hipLaunchKernelGGL(flip_transpose_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w, wt, M, C, KH, KW);
  PRN_CHECK_LAUNCH("prn_weight_flip_transpose");
  return 0;
}

extern "C" int prn_pad_fold(const float* dp, float* dx, int B, int C, int H, int W, int up2, void* stream) {
  PRN_REQUIRE(dp && dx && B > 0 && C > 0 && H > 1 && W > 1, "prn_pad_fold: bad arguments");
  const int64_t n = (int64_t)B * C * H * W;
  hipLaunchKernelGGL(pad_fold_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dp, dx, B * C, H, W, up2);
  PRN_CHECK_LAUNCH("prn_pad_fold");
  return 0;
}

extern "C" int prn_channel_sum(const float* x, float* out, int B, int C, int HW, void* stream) {
  PRN_REQUIRE(x && out && B > 0 && C > 0 && HW > 0, "prn_channel_sum: bad arguments");
  hipLaunchKernelGGL(channel_sum_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, out, B, C, HW);
  PRN_CHECK_LAUNCH("prn_channel_sum");
  return 0;
}

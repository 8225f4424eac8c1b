// Winograd F(4x4, 3x3) path for the stride-1, pad-1 3x3 convolutions (reference call sites: the bottleneck conv2 of
// models/backbone.py, the FPN smoothing convs of models/fpn.py, the mask / depth heads of planerecnet.py) and for their
// input gradients (same kernel run over dy with the 180-degree-rotated weights).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A        d: 6x6 input patch, g: 3x3 taps, Y: 4x4 outputs   (Lavin & Gray 2016)
//
// 36 multiplies per 16 outputs instead of 144: the element-wise product summed over input channels is 36 independent GEMMs
// U_z[M x C] * V_z[C x P] (P = tiles in the batch) that run on the MFMA kernel of prn_conv.hip (prn_gemm_batched); the
// three transforms here are HBM-bound streaming kernels:
//   winograd_input_kernel   x [B,C,H,W]        -> V [36][C][P]     reads 1 float4 + 2 scalars per patch row, writes coalesced along P
//   winograd_output_kernel  Y' [36][M][P]      -> y [B,M,H,W]      (+ bias, + addend, ReLU) float4 rows
//   winograd_weights_kernel w [M,C,3,3]        -> U [36][M][C] and / or the dgrad operand U' [36][C][M], batched over a model
// fp32 throughout; the transform constants are exact in binary except the 1/6, 1/12, 1/24 of G (relative error of the
// result ~1e-6, covered by the parity tests against the direct kernel and the CPU oracle).
#include "prn_common.h"

namespace {

__device__ __forceinline__ int reflect1(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * n - 2 - i : i;
}

// Tile geometry of a (possibly ragged) batch: segment t holds B maps of H[t] x W[t] stored [B][channels][H][W] back to back
// (include/prn.h: prn_ragged); its 4x4 tiles occupy columns p0[t] .. p0[t+1] of the transform-domain operands and its maps
// start hw0[t] * B * channels elements into the packed tensor.  A dense tensor is the one-segment case.
struct WSeg {
  int nseg, B;
  int p0[PRN_MAX_SEGMENTS + 1], hw0[PRN_MAX_SEGMENTS], H[PRN_MAX_SEGMENTS], W[PRN_MAX_SEGMENTS];
};
struct TilePos { int H, W, b, ty, tx; size_t base; };       // base: element offset of map (b, channel 0) -> add channel * H * W
// (static indices only: indexing the by-value kernel argument dynamically sends it to scratch, see prn_conv.hip)
__device__ __forceinline__ TilePos locate(const WSeg& g, int p, int channels) {
  int H = g.H[0], W = g.W[0], q0 = 0, hw0 = 0;
#pragma unroll
  for (int t = 1; t < PRN_MAX_SEGMENTS; ++t)
    if (t < g.nseg && p >= g.p0[t]) { H = g.H[t]; W = g.W[t]; q0 = g.p0[t]; hw0 = g.hw0[t]; }
  const int TW = W >> 2, TH = (H + 3) >> 2, q = p - q0;
  TilePos r;
  r.H = H; r.W = W;
  r.tx = q % TW; r.ty = (q / TW) % TH; r.b = q / (TW * TH);
  r.base = ((size_t)hw0 * g.B + (size_t)r.b * H * W) * channels;
  return r;
}

// (bt6 / at6, the B^T and A^T passes of the transforms: prn_common.h -- the one-launch BatchNorm kernels of prn_norm.hip use bt6 too)
// o = G g
__device__ __forceinline__ void g6(float g0, float g1, float g2, float* o) {
  o[0] = 0.25f * g0;
  o[1] = -(g0 + g1 + g2) * (1.f / 6.f);
  o[2] = -(g0 - g1 + g2) * (1.f / 6.f);
  o[3] = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
  o[4] = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
  o[5] = g2;
}

// One thread = one 6x6 patch of one channel.  grid (ceil(P/256), C); W % 4 == 0, so the four interior columns of every
// patch row are one aligned float4 and only the two halo columns are scalar loads (which hit the neighbours' lines).
template <int MODE>
__global__ __launch_bounds__(256) void winograd_input_kernel(const float* __restrict__ x, float* __restrict__ V, int C, int P, int P4, WSeg g) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int c = blockIdx.y;
  const TilePos tp = locate(g, p, C);
  const int H = tp.H, W = tp.W, ty = tp.ty;
  const int w0 = 4 * tp.tx;
  float d[6][6];
  if (MODE == PRN_IN_EMBED1) {
    // (H, W) is a virtual zero tensor; the real one, [B][C][H-2][W-4], sits at (1, 1) inside it (dense batches only)
    const int Hr = H - 2, Wr = W - 4;
    const float* xc = x + ((size_t)tp.b * C + c) * Hr * Wr;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int rr = 4 * ty - 2 + i;
      const bool okr = (unsigned)rr < (unsigned)Hr;
      const float* row = xc + (size_t)(okr ? rr : 0) * Wr;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int rc = w0 - 2 + j;
        const bool ok = okr && (unsigned)rc < (unsigned)Wr;
        const float v = row[ok ? rc : 0];
        d[i][j] = ok ? v : 0.f;
      }
    }
  } else {
  const float* xc = x + tp.base + (size_t)c * H * W;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    int ih = 4 * ty - 1 + i;
    bool ok = true;
    if (MODE == PRN_IN_REFLECT) ih = reflect1(ih, H); else ok = (unsigned)ih < (unsigned)H;
    ih = ok ? ih : 0;
    const float* row = xc + (size_t)ih * W;
    const float4 q = *reinterpret_cast<const float4*>(row + w0);
    int il = w0 - 1, ir = w0 + 4;
    bool okl = il >= 0, okr = ir < W;
    if (MODE == PRN_IN_REFLECT) { il = okl ? il : 1; ir = okr ? ir : W - 2; okl = okr = true; }
    const float l = row[okl ? il : 0], r = row[okr ? ir : 0];
    d[i][0] = (ok && okl) ? l : 0.f;
    d[i][1] = ok ? q.x : 0.f; d[i][2] = ok ? q.y : 0.f; d[i][3] = ok ? q.z : 0.f; d[i][4] = ok ? q.w : 0.f;
    d[i][5] = (ok && okr) ? r : 0.f;
  }
  }
  float t[6][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {                 // columns: t = B^T d
    float v[6], o[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = d[i][j];
    bt6(v, o);
#pragma unroll
    for (int i = 0; i < 6; ++i) t[i][j] = o[i];
  }
  float* out = V + (size_t)c * P4 + p;
  const size_t zs = (size_t)C * P4;
#pragma unroll
  for (int i = 0; i < 6; ++i) {                 // rows: V = t B
    float o[6];
    bt6(t[i], o);
#pragma unroll
    for (int j = 0; j < 6; ++j) out[(size_t)(i * 6 + j) * zs] = o[j];
  }
}

// One thread = one 4x4 output tile of one channel.  grid (ceil(P/256), M).
__global__ __launch_bounds__(256) void winograd_output_kernel(const float* __restrict__ Y, const float* __restrict__ bias, const float* __restrict__ addend,
                                                              float* __restrict__ y, int M, int P, int P4, int relu, WSeg g) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int m = blockIdx.y;
  const TilePos tp = locate(g, p, M);
  const int H = tp.H, W = tp.W, ty = tp.ty, tx = tp.tx;
  const float* in = Y + (size_t)m * P4 + p;
  const size_t zs = (size_t)M * P4;
  float t[4][6];                                // t = A^T Y'  (4 x 6)
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float v[6], o[4];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = in[(size_t)(i * 6 + j) * zs];
    at6(v, o);
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i][j] = o[i];
  }
  const float bv = bias ? bias[m] : 0.f;
  const size_t plane = tp.base + (size_t)m * H * W;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int oh = 4 * ty + i;
    if (oh >= H) break;
    float o[4];
    at6(t[i], o);
    const size_t idx = plane + (size_t)oh * W + 4 * tx;
    float4 r = make_float4(o[0] + bv, o[1] + bv, o[2] + bv, o[3] + bv);
    if (addend) {
      const float4 q = *reinterpret_cast<const float4*>(addend + idx);
      r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
    }
    if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
    *reinterpret_cast<float4*>(y + idx) = r;
  }
}

// ---- output transform + BatchNorm in one launch (small maps) ---------------------------------------------------------------------
// The layers of the backbone's stages 3 / 4 have <= 12288 values per channel: ONE workgroup owns a channel, transforms all of its tiles
// (<= 768: three per thread) and keeps the 4x4 outputs in registers, so what follows the 3x3 convolution in the reference -- BatchNorm in
// training mode (models/backbone.py:60-62), or, in the backward pass, the backward of the BatchNorm + ReLU in front of it (:57-58) --
// needs no second kernel and no second pass over the tensor.  Dense batches only; the transform arithmetic is winograd_output_kernel's.
template <int NT>
__device__ __forceinline__ void wino_tiles_out(const float* __restrict__ Y, int m, int M, int P, int P4, int H, int W, float4 (&o)[NT][4], size_t (&off)[NT],
                                               int (&rows)[NT]) {
  const int TW = W >> 2, TH = (H + 3) >> 2;
  const size_t zs = (size_t)M * P4;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int p = threadIdx.x + t * 256;
    const bool live = p < P;
    const int q = live ? p : 0;
    const int tx = q % TW, ty = (q / TW) % TH, b = q / (TW * TH);
    const float* in = Y + (size_t)m * P4 + q;
    float v[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) v[k] = in[(size_t)k * zs];          // (unconditional: a dead slot re-reads tile 0)
    float tt[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float c6[6], r4[4];
#pragma unroll
      for (int i = 0; i < 6; ++i) c6[i] = v[i * 6 + j];
      at6(c6, r4);
#pragma unroll
      for (int i = 0; i < 4; ++i) tt[i][j] = r4[i];
    }
    rows[t] = live ? min(4, H - 4 * ty) : 0;                         // valid rows of the tile (the last tile row may hang over the map)
    off[t] = (((size_t)b * M + m) * H + 4 * ty) * W + 4 * tx;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float r4[4];
      at6(tt[i], r4);
      o[t][i] = i < rows[t] ? make_float4(r4[0], r4[1], r4[2], r4[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

template <int NT>
__global__ __launch_bounds__(256) void winograd_output_bn_fwd_kernel(const float* __restrict__ Y, float* __restrict__ xout, float* __restrict__ stats,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ z,
                                                                     float* __restrict__ rmean, float* __restrict__ rvar, int B, int M, int P, int P4, int H, int W,
                                                                     float eps, float momentum, int relu) {
  const int m = blockIdx.x;
  float4 o[NT][4];
  size_t off[NT];
  int rows[NT];
  wino_tiles_out<NT>(Y, m, M, P, P4, H, W, o, off, rows);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                    // (rows outside the map are zero)
      const float4 v = o[t][i];
      s1 += (v.x + v.y) + (v.z + v.w);
      s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
  __shared__ double sm[16];
  double d1 = s1, d2 = s2;
  block_sum2(d1, d2, sm);
  const double count = (double)B * H * W, mean = d1 / count;
  double var = d2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float mean_f = (float)mean, istd_f = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0) {
    stats[m] = mean_f;
    stats[M + m] = istd_f;
    if (rmean) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      rmean[m] = (1.f - momentum) * rmean[m] + momentum * mean_f;
      rvar[m] = (1.f - momentum) * rvar[m] + momentum * (float)unbiased;
    }
  }
  const float sc = istd_f * gamma[m], sh = beta[m] - mean_f * sc;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i >= rows[t]) continue;
      const float4 v = o[t][i];
      *reinterpret_cast<float4*>(xout + off[t] + (size_t)i * W) = v;
      float4 r;
      r.x = fmaf(v.x, sc, sh); r.y = fmaf(v.y, sc, sh); r.z = fmaf(v.z, sc, sh); r.w = fmaf(v.w, sc, sh);
      if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
      *reinterpret_cast<float4*>(z + off[t] + (size_t)i * W) = r;
    }
}

// backward of BatchNorm (+ ReLU, mask recomputed from x as bn_small_bwd_kernel's relu == 2) whose output gradient is the result of a Winograd
// input-gradient convolution still in the transform domain: dx = gamma * istd * (g - mean(g) - xhat * mean(g * xhat))
template <int NT>
__global__ __launch_bounds__(256) void winograd_output_bn_bwd_kernel(const float* __restrict__ Y, const float* __restrict__ x, const float* __restrict__ stats,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ dx,
                                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int M, int P, int P4, int H,
                                                                     int W, int relu) {
  const int m = blockIdx.x;
  float4 g[NT][4];
  size_t off[NT];
  int rows[NT];
  wino_tiles_out<NT>(Y, m, M, P, P4, H, W, g, off, rows);
  const float mean = stats[m], istd = stats[M + m], gi = gamma[m] * istd;
  const float sc = relu ? gi : 0.f, sh = relu ? beta[m] - mean * sc : 0.f;
  float4 xv[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      xv[t][i] = *reinterpret_cast<const float4*>(x + (i < rows[t] ? off[t] + (size_t)i * W : 0));      // (dead rows re-read element 0; their g is zero)
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4& q = g[t][i];
      const float4 v = xv[t][i];
      if (relu) {
        q.x = fmaf(v.x, sc, sh) > 0.f ? q.x : 0.f; q.y = fmaf(v.y, sc, sh) > 0.f ? q.y : 0.f;
        q.z = fmaf(v.z, sc, sh) > 0.f ? q.z : 0.f; q.w = fmaf(v.w, sc, sh) > 0.f ? q.w : 0.f;
      }
      s1 += (q.x + q.y) + (q.z + q.w);
      s2 += (q.x * (v.x - mean) + q.y * (v.y - mean)) + (q.z * (v.z - mean) + q.w * (v.w - mean));
    }
  s2 *= istd;
  __shared__ double sm[16];
  double t1 = s1, t2 = s2;
  block_sum2(t1, t2, sm);
  if (threadIdx.x == 0) {
    if (dbeta) dbeta[m] = (float)t1;
    if (dgamma) dgamma[m] = (float)t2;
  }
  const float inv_count = 1.f / ((float)B * H * W);
  const float m1 = (float)t1 * inv_count, m2 = (float)t2 * inv_count;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i >= rows[t]) continue;
      const float4 q = g[t][i], v = xv[t][i];
      float4 r;
      r.x = gi * (q.x - m1 - (v.x - mean) * istd * m2); r.y = gi * (q.y - m1 - (v.y - mean) * istd * m2);
      r.z = gi * (q.z - m1 - (v.z - mean) * istd * m2); r.w = gi * (q.w - m1 - (v.w - mean) * istd * m2);
      *reinterpret_cast<float4*>(dx + off[t] + (size_t)i * W) = r;
    }
}

// One workgroup = one 32 x 32 (m, c) block of one weight tensor (items are located by bisection over `first`, as in
// flip_transpose_batched_kernel).  Pass A writes U[z][m][c] with c along the lanes; pass B re-reads the same 36 KiB block
// (L1/L2 hits) with m along the lanes and writes the rotated-tap transform U'[z][c][m] -- both stores coalesced.
__device__ __forceinline__ void transform_taps(const float* g, bool rot, float* u /*36*/) {
  float k[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) k[i] = rot ? g[8 - i] : g[i];
  float t[6][3];                                // t = G g   (6 x 3)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float o[6];
    g6(k[j], k[3 + j], k[6 + j], o);
#pragma unroll
    for (int i = 0; i < 6; ++i) t[i][j] = o[i];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) g6(t[i][0], t[i][1], t[i][2], u + i * 6);     // rows: u = t G^T
}

__global__ __launch_bounds__(256) void winograd_weights_kernel(const prn_winograd_item* __restrict__ items, int n, int64_t total) {
  const int64_t blk = blockIdx.x;
  if (blk >= total) return;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].first <= blk) lo = mid; else hi = mid - 1;
  }
  const prn_winograd_item it = items[lo];
  const int tilesC = (it.C + 31) / 32;
  const int t = (int)(blk - it.first), m0 = (t / tilesC) * 32, c0 = (t % tilesC) * 32;
  const size_t zs = (size_t)it.M * it.C;
  for (int i = threadIdx.x; i < 1024; i += 256) {
    if (it.u) {
      const int m = m0 + (i >> 5), c = c0 + (i & 31);
      if (m < it.M && c < it.C) {
        float u[36];
        transform_taps(it.src + ((size_t)m * it.C + c) * 9, false, u);
#pragma unroll
        for (int z = 0; z < 36; ++z) it.u[z * zs + (size_t)m * it.C + c] = u[z];
      }
    }
    if (it.ut) {
      const int c = c0 + (i >> 5), m = m0 + (i & 31);
      if (m < it.M && c < it.C) {
        float u[36];
        transform_taps(it.src + ((size_t)m * it.C + c) * 9, true, u);
#pragma unroll
        for (int z = 0; z < 36; ++z) it.ut[z * zs + (size_t)c * it.M + m] = u[z];
      }
    }
  }
}

// o = A v   (4 -> 6): the transpose of at6, used by the weight gradient (dY' = A dy A^T)
__device__ __forceinline__ void a4(const float* v, float* o) {
  o[0] = v[0];
  o[1] = v[0] + v[1] + v[2] + v[3];
  o[2] = v[0] - v[1] + v[2] - v[3];
  o[3] = v[0] + 2.f * v[1] + 4.f * v[2] + 8.f * v[3];
  o[4] = v[0] - 2.f * v[1] + 4.f * v[2] - 8.f * v[3];
  o[5] = v[3];
}
// o = G^T v   (6 -> 3)
__device__ __forceinline__ void gt6(const float* v, float* o) {
  const float s12 = v[1] + v[2], s34 = v[3] + v[4];
  o[0] = 0.25f * v[0] - s12 * (1.f / 6.f) + s34 * (1.f / 24.f);
  o[1] = (v[2] - v[1]) * (1.f / 6.f) + (v[3] - v[4]) * (1.f / 12.f);
  o[2] = -s12 * (1.f / 6.f) + s34 * (1.f / 6.f) + v[5];
}

// Weight gradient, step 1: dy [B,M,H,W] -> dY' [36][M][P], dY' = A dy A^T per 4x4 tile (rows beyond H are zero).
__global__ __launch_bounds__(256) void winograd_dy_kernel(const float* __restrict__ dy, float* __restrict__ Y, int M, int P, int P4, WSeg g) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int m = blockIdx.y;
  const TilePos tp = locate(g, p, M);
  const int H = tp.H, W = tp.W, ty = tp.ty;
  const float* src = dy + tp.base + (size_t)m * H * W + 4 * tp.tx;
  float d[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int oh = 4 * ty + i;
    const bool ok = oh < H;
    const float4 q = *reinterpret_cast<const float4*>(src + (size_t)(ok ? oh : 0) * W);
    d[i][0] = ok ? q.x : 0.f; d[i][1] = ok ? q.y : 0.f; d[i][2] = ok ? q.z : 0.f; d[i][3] = ok ? q.w : 0.f;
  }
  float t[6][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v[4], o[6];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = d[i][j];
    a4(v, o);
#pragma unroll
    for (int i = 0; i < 6; ++i) t[i][j] = o[i];
  }
  float* out = Y + (size_t)m * P4 + p;
  const size_t zs = (size_t)M * P4;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float o[6];
    a4(t[i], o);
#pragma unroll
    for (int j = 0; j < 6; ++j) out[(size_t)(i * 6 + j) * zs] = o[j];
  }
}

// Weight gradient, step 3: partial dU [splits][36][M][C] -> dw [M][C][3][3] = G^T (sum over splits) G.
// A workgroup owns 64 (m, c) pairs; its four waves each sum every fourth split (coalesced along the pairs), the four partial
// sums meet in LDS in fixed order (deterministic), and wave 0 applies the transform.
__global__ __launch_bounds__(256) void winograd_dw_kernel(const float* __restrict__ part, float* __restrict__ dw, int M, int C, int splits) {
  __shared__ float sm[3][36][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t MC = (int64_t)M * C, i = (int64_t)blockIdx.x * 64 + lane;
  float u[36];
#pragma unroll
  for (int z = 0; z < 36; ++z) u[z] = 0.f;
  if (i < MC)
    for (int k = g; k < splits; k += 4) {
#pragma unroll
      for (int z = 0; z < 36; ++z) u[z] += part[((size_t)k * 36 + z) * MC + i];
    }
  if (g > 0) {
#pragma unroll
    for (int z = 0; z < 36; ++z) sm[g - 1][z][lane] = u[z];
  }
  __syncthreads();
  if (g > 0 || i >= MC) return;
#pragma unroll
  for (int z = 0; z < 36; ++z) u[z] = ((u[z] + sm[0][z][lane]) + sm[1][z][lane]) + sm[2][z][lane];
  float t[3][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float v[6], o[3];
#pragma unroll
    for (int a = 0; a < 6; ++a) v[a] = u[a * 6 + j];
    gt6(v, o);
#pragma unroll
    for (int a = 0; a < 3; ++a) t[a][j] = o[a];
  }
  float* out = dw + (size_t)i * 9;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float o[3];
    gt6(t[a], o);
    out[a * 3 + 0] = o[0]; out[a * 3 + 1] = o[1]; out[a * 3 + 2] = o[2];
  }
}

}  // namespace

static inline int64_t pad4(int64_t p) { return (p + 3) & ~(int64_t)3; }

// Fills the segment table; returns the tile count P (unpadded) or -1 after prn_set_error.
static int64_t make_seg(WSeg& g, const prn_ragged* rg, int B, int H, int W, const char* who) {
  prn_ragged one;
  if (rg == nullptr) { one.nseg = 1; one.H[0] = H; one.W[0] = W; rg = &one; }
  if (!(rg->nseg >= 1 && rg->nseg <= PRN_MAX_SEGMENTS && B > 0)) { prn_set_error("%s: bad segment table", who); return -1; }
  g.nseg = rg->nseg; g.B = B;
  int64_t p = 0, hw = 0;
  for (int t = 0; t < PRN_MAX_SEGMENTS; ++t) {
    const bool live = t < rg->nseg;
    const int h = live ? rg->H[t] : 0, w = live ? rg->W[t] : 0;
    if (live && !(h >= 5 && w >= 4 && (w & 3) == 0)) { prn_set_error("%s: W %% 4 == 0 and H >= 5 required (segment %d: H=%d W=%d)", who, t, h, w); return -1; }
    g.p0[t] = (int)p; g.hw0[t] = (int)hw; g.H[t] = h; g.W[t] = w;
    p += (int64_t)B * ((h + 3) / 4) * (w / 4);
    hw += (int64_t)h * w;
    if (live && (p >= (1LL << 30) || hw * B >= (1LL << 30))) { prn_set_error("%s: too many tiles", who); return -1; }
  }
  g.p0[PRN_MAX_SEGMENTS] = (int)p;
  return p;
}

extern "C" int64_t prn_winograd_tiles(int B, int H, int W) {
  WSeg g;
  const int64_t P = make_seg(g, nullptr, B, H, W, "prn_winograd_tiles");
  return P < 0 ? -1 : pad4(P);
}
extern "C" int64_t prn_winograd_tiles_ragged(const prn_ragged* rg, int B) {
  WSeg g;
  const int64_t P = rg ? make_seg(g, rg, B, 0, 0, "prn_winograd_tiles_ragged") : -1;
  return P < 0 ? -1 : pad4(P);
}

extern "C" int prn_winograd_weights_batched(const prn_winograd_item* items_dev, int n_items, int64_t total_blocks, void* stream) {
  PRN_REQUIRE(items_dev && n_items > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "prn_winograd_weights_batched: bad arguments");
  hipLaunchKernelGGL(winograd_weights_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, n_items, total_blocks);
  PRN_CHECK_LAUNCH("prn_winograd_weights_batched");
  return 0;
}

namespace {
int input_impl(const float* x, float* V, const prn_ragged* rg, int B, int C, int H, int W, int in_mode, void* stream) {
  PRN_REQUIRE(x && V && C > 0 && C < 65536, "prn_winograd_input: bad arguments");
  PRN_REQUIRE(in_mode == PRN_IN_ZERO || in_mode == PRN_IN_REFLECT || (in_mode == PRN_IN_EMBED1 && rg == nullptr && H > 2 && W > 4),
              "prn_winograd_input: zero / reflect padding, or PRN_IN_EMBED1 on a dense batch");
  PRN_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "prn_winograd_input: x must be 16-byte aligned");
  WSeg g;
  const int64_t P = make_seg(g, rg, B, H, W, "prn_winograd_input");
  if (P < 0) return 2;
  const dim3 grid(cdiv(P, 256), C), block(256);
  if (in_mode == PRN_IN_ZERO) hipLaunchKernelGGL((winograd_input_kernel<PRN_IN_ZERO>), grid, block, 0, (hipStream_t)stream, x, V, C, (int)P, (int)pad4(P), g);
  else if (in_mode == PRN_IN_EMBED1) hipLaunchKernelGGL((winograd_input_kernel<PRN_IN_EMBED1>), grid, block, 0, (hipStream_t)stream, x, V, C, (int)P, (int)pad4(P), g);
  else hipLaunchKernelGGL((winograd_input_kernel<PRN_IN_REFLECT>), grid, block, 0, (hipStream_t)stream, x, V, C, (int)P, (int)pad4(P), g);
  PRN_CHECK_LAUNCH("prn_winograd_input");
  return 0;
}

int output_impl(const float* Y, const float* bias, const float* addend, float* y, const prn_ragged* rg, int B, int M, int H, int W, int epilogue, void* stream) {
  PRN_REQUIRE(Y && y && M > 0 && M < 65536, "prn_winograd_output: bad arguments");
  PRN_REQUIRE(epilogue == PRN_EPI_NONE || epilogue == PRN_EPI_RELU, "prn_winograd_output: epilogue none or ReLU only");
  PRN_REQUIRE((reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(addend) & 15) == 0, "prn_winograd_output: y / addend must be 16-byte aligned");
  WSeg g;
  const int64_t P = make_seg(g, rg, B, H, W, "prn_winograd_output");
  if (P < 0) return 2;
  hipLaunchKernelGGL(winograd_output_kernel, dim3(cdiv(P, 256), M), dim3(256), 0, (hipStream_t)stream, Y, bias, addend, y, M, (int)P, (int)pad4(P),
                     epilogue == PRN_EPI_RELU ? 1 : 0, g);
  PRN_CHECK_LAUNCH("prn_winograd_output");
  return 0;
}

int dy_impl(const float* dy, float* Y, const prn_ragged* rg, int B, int M, int H, int W, void* stream) {
  PRN_REQUIRE(dy && Y && M > 0 && M < 65536, "prn_winograd_dy: bad arguments");
  PRN_REQUIRE((reinterpret_cast<uintptr_t>(dy) & 15) == 0, "prn_winograd_dy: dy must be 16-byte aligned");
  WSeg g;
  const int64_t P = make_seg(g, rg, B, H, W, "prn_winograd_dy");
  if (P < 0) return 2;
  hipLaunchKernelGGL(winograd_dy_kernel, dim3(cdiv(P, 256), M), dim3(256), 0, (hipStream_t)stream, dy, Y, M, (int)P, (int)pad4(P), g);
  PRN_CHECK_LAUNCH("prn_winograd_dy");
  return 0;
}

int64_t fwd_ws(const prn_ragged* rg, int B, int C, int H, int W, int M, const prn_gemm_opts* opts) {
  WSeg g;
  const int64_t P = make_seg(g, rg, B, H, W, "prn_conv3x3_winograd_ws_bytes");
  if (P < 0 || C <= 0 || M <= 0) return -1;
  const int64_t P4 = pad4(P);
  const int64_t gb = prn_gemm_batched_ws_bytes(M, C, (int)P4, 36, opts);
  return gb < 0 ? -1 : ((4 * 36 * (int64_t)(C + M) * P4 + 255) & ~255LL) + gb;
}

int fwd_impl(const float* x, const float* U, const void* u_images, const float* bias, const float* addend, float* y, void* ws, const prn_ragged* rg, int B, int C,
             int H, int W, int M, int in_mode, int epilogue, const prn_gemm_opts* opts, void* stream) {
  PRN_REQUIRE(ws && U, "prn_conv3x3_winograd: workspace and transformed weights required");
  WSeg g;
  const int64_t P = make_seg(g, rg, B, H, W, "prn_conv3x3_winograd");
  if (P < 0) return 2;
  const int64_t P4 = pad4(P);
  float* V = (float*)ws;
  float* Yt = V + 36 * (int64_t)C * P4;
  if (int e = input_impl(x, V, rg, B, C, H, W, in_mode, stream)) return e;
  char* gws = (char*)ws + ((4 * 36 * (int64_t)(C + M) * P4 + 255) & ~255LL);      // (only touched when the split kernel cuts U itself)
  if (int e = prn_gemm_batched(M, C, (int)P4, 36, U, u_images, V, Yt, prn_gemm_batched_ws_bytes(M, C, (int)P4, 36, opts) > 0 ? gws : nullptr, opts, stream)) return e;
  return output_impl(Yt, bias, addend, y, rg, B, M, H, W, epilogue, stream);
}

int64_t wgrad_ws(const prn_ragged* rg, int B, int C, int H, int W, int M, const prn_gemm_opts* opts) {
  WSeg g;
  const int64_t P = make_seg(g, rg, B, H, W, "prn_winograd_wgrad_ws_bytes");
  if (P < 0) return -1;
  const int64_t P4 = pad4(P);
  const int splits = prn_gemm_batched_nt_splits(M, C, (int)P4, 36, opts);
  return 4 * (36 * (int64_t)(C + M) * P4 + (int64_t)splits * 36 * M * C);
}

// phase: 0 = everything, 1 = transforms of x and dy, 2 = the 36 products, 3 = reduction + G^T . G (profilers bracket them separately)
int wgrad_impl(const float* x, const float* dy, float* dw, void* ws, const prn_ragged* rg, int B, int C, int H, int W, int M, int in_mode, const prn_gemm_opts* opts,
               void* stream, int phase, const float* V_in = nullptr) {
  PRN_REQUIRE(ws && (x || V_in) && dy && dw, "prn_conv3x3_winograd_wgrad: null tensor / workspace");
  WSeg g;
  const int64_t P = make_seg(g, rg, B, H, W, "prn_conv3x3_winograd_wgrad");
  if (P < 0) return 2;
  const int64_t P4 = pad4(P);
  float* V = V_in ? const_cast<float*>(V_in) : (float*)ws;      // V_in: B^T x B kept from the forward pass of the same layer
  float* Yt = (float*)ws + 36 * (int64_t)C * P4;
  float* part = Yt + 36 * (int64_t)M * P4;
  if (phase == 0 || phase == 1) {
    if (!V_in)
      if (int e = input_impl(x, V, rg, B, C, H, W, in_mode, stream)) return e;
    if (int e = dy_impl(dy, Yt, rg, B, M, H, W, stream)) return e;
    if (P4 != P) {                                     // the products reduce over P4 columns: the padding must be zero
      // (at most 3 columns per row: cleared with one strided memset per operand; a kept V has finite padding only
      // if its producer cleared it, so it is cleared here as well)
      const hipError_t e1 = hipMemset2DAsync(V + P, P4 * 4, 0, (P4 - P) * 4, 36 * (size_t)C, (hipStream_t)stream);
      const hipError_t e2 = hipMemset2DAsync(Yt + P, P4 * 4, 0, (P4 - P) * 4, 36 * (size_t)M, (hipStream_t)stream);
      PRN_REQUIRE(e1 == hipSuccess && e2 == hipSuccess, "prn_conv3x3_winograd_wgrad: clearing the tile padding failed: %s",
                  hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    }
  }
  if (phase == 0 || phase == 2)
    if (int e = prn_gemm_batched_nt(M, C, (int)P4, 36, Yt, V, part, opts, stream)) return e;
  if (phase == 0 || phase == 3)
    return prn_winograd_dw(part, dw, M, C, prn_gemm_batched_nt_splits(M, C, (int)P4, 36, opts), stream);
  return 0;
}
}  // namespace

extern "C" int prn_winograd_input(const float* x, float* V, int B, int C, int H, int W, int in_mode, void* stream) {
  return input_impl(x, V, nullptr, B, C, H, W, in_mode, stream);
}
extern "C" int prn_winograd_output(const float* Y, const float* bias, const float* addend, float* y, int B, int M, int H, int W, int epilogue, void* stream) {
  return output_impl(Y, bias, addend, y, nullptr, B, M, H, W, epilogue, stream);
}
extern "C" int prn_winograd_dy(const float* dy, float* Y, int B, int M, int H, int W, void* stream) {
  return dy_impl(dy, Y, nullptr, B, M, H, W, stream);
}
// Output transform of Y' [36][M][P] fused with what consumes the convolution's result when a whole channel fits one workgroup's registers
// (prn_bn_kernel_kind(B, H * W) == 1, i.e. <= 768 tiles): see winograd_output_bn_fwd_kernel / _bwd_kernel.
extern "C" int prn_winograd_output_bn_fwd(const float* Yt, float* x_out, float* stats, const float* gamma, const float* beta, float* y, float* running_mean,
                                          float* running_var, int B, int M, int H, int W, float eps, float momentum, int relu, void* stream) {
  PRN_REQUIRE(Yt && x_out && stats && gamma && beta && y && M > 0 && M < 65536, "prn_winograd_output_bn_fwd: bad arguments");
  PRN_REQUIRE(((reinterpret_cast<uintptr_t>(x_out) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, "prn_winograd_output_bn_fwd: tensors must be 16-byte aligned");
  WSeg g;
  const int64_t P = make_seg(g, nullptr, B, H, W, "prn_winograd_output_bn_fwd");
  if (P < 0) return 2;
  PRN_REQUIRE(prn_bn_kernel_kind(B, H * W) == 1 && P <= 768, "prn_winograd_output_bn_fwd: only for maps the one-launch BatchNorm kernels take");
  hipStream_t st = (hipStream_t)stream;
#define PRN_WOB(NT_) hipLaunchKernelGGL((winograd_output_bn_fwd_kernel<NT_>), dim3(M), dim3(256), 0, st, Yt, x_out, stats, gamma, beta, y, running_mean, running_var, \
                                        B, M, (int)P, (int)pad4(P), H, W, eps, momentum, relu)
  if (P <= 256) PRN_WOB(1); else if (P <= 512) PRN_WOB(2); else PRN_WOB(3);
#undef PRN_WOB
  PRN_CHECK_LAUNCH("prn_winograd_output_bn_fwd");
  return 0;
}
extern "C" int prn_winograd_output_bn_bwd(const float* Yt, const float* x, const float* stats, const float* gamma, const float* beta, float* dx, float* dgamma,
                                          float* dbeta, int B, int M, int H, int W, int relu, void* stream) {
  PRN_REQUIRE(Yt && x && stats && gamma && dx && (beta || !relu) && M > 0 && M < 65536, "prn_winograd_output_bn_bwd: bad arguments");
  PRN_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0, "prn_winograd_output_bn_bwd: tensors must be 16-byte aligned");
  WSeg g;
  const int64_t P = make_seg(g, nullptr, B, H, W, "prn_winograd_output_bn_bwd");
  if (P < 0) return 2;
  PRN_REQUIRE(prn_bn_kernel_kind(B, H * W) == 1 && P <= 768, "prn_winograd_output_bn_bwd: only for maps the one-launch BatchNorm kernels take");
  hipStream_t st = (hipStream_t)stream;
#define PRN_WOB(NT_) hipLaunchKernelGGL((winograd_output_bn_bwd_kernel<NT_>), dim3(M), dim3(256), 0, st, Yt, x, stats, gamma, beta, dx, dgamma, dbeta, B, M, (int)P, \
                                        (int)pad4(P), H, W, relu)
  if (P <= 256) PRN_WOB(1); else if (P <= 512) PRN_WOB(2); else PRN_WOB(3);
#undef PRN_WOB
  PRN_CHECK_LAUNCH("prn_winograd_output_bn_bwd");
  return 0;
}
extern "C" int prn_winograd_dw(const float* partials, float* dw, int M, int C, int splits, void* stream) {
  PRN_REQUIRE(partials && dw && M > 0 && C > 0 && splits > 0, "prn_winograd_dw: bad arguments");
  hipLaunchKernelGGL(winograd_dw_kernel, dim3(cdiv((int64_t)M * C, 64)), dim3(256), 0, (hipStream_t)stream, partials, dw, M, C, splits);
  PRN_CHECK_LAUNCH("prn_winograd_dw");
  return 0;
}

// x -> V -> (36 GEMMs) -> Y' -> y in one call.  ws: prn_conv3x3_winograd_ws_bytes.
extern "C" int64_t prn_conv3x3_winograd_ws_bytes(int B, int C, int H, int W, int M, const prn_gemm_opts* opts) { return fwd_ws(nullptr, B, C, H, W, M, opts); }
extern "C" int64_t prn_conv3x3_winograd_ragged_ws_bytes(const prn_ragged* rg, int B, int C, int M, const prn_gemm_opts* opts) {
  return rg ? fwd_ws(rg, B, C, 0, 0, M, opts) : -1;
}
extern "C" int prn_conv3x3_winograd(const float* x, const float* U, const void* u_images, const float* bias, const float* addend, float* y, void* ws, int B, int C,
                                    int H, int W, int M, int in_mode, int epilogue, const prn_gemm_opts* opts, void* stream) {
  return fwd_impl(x, U, u_images, bias, addend, y, ws, nullptr, B, C, H, W, M, in_mode, epilogue, opts, stream);
}
// The same over a ragged batch (prn_ragged: every segment convolved with the same weights; zero padding).
extern "C" int prn_conv3x3_winograd_ragged(const float* x, const float* U, const void* u_images, const float* bias, const float* addend, float* y, void* ws,
                                           const prn_ragged* rg, int B, int C, int M, int epilogue, const prn_gemm_opts* opts, void* stream) {
  PRN_REQUIRE(rg, "prn_conv3x3_winograd_ragged: null segment table");
  return fwd_impl(x, U, u_images, bias, addend, y, ws, rg, B, C, 0, 0, M, PRN_IN_ZERO, epilogue, opts, stream);
}

/* ---- weight gradient: dw = G^T [ sum_tiles (A dy A^T) .* (B^T x B) ] G */
extern "C" int64_t prn_winograd_wgrad_ws_bytes(int B, int C, int H, int W, int M, const prn_gemm_opts* opts) { return wgrad_ws(nullptr, B, C, H, W, M, opts); }
extern "C" int64_t prn_winograd_wgrad_ragged_ws_bytes(const prn_ragged* rg, int B, int C, int M, const prn_gemm_opts* opts) {
  return rg ? wgrad_ws(rg, B, C, 0, 0, M, opts) : -1;
}
extern "C" int prn_conv3x3_winograd_wgrad(const float* x, const float* dy, float* dw, void* ws, int B, int C, int H, int W, int M, int in_mode,
                                          const prn_gemm_opts* opts, void* stream, int phase) {
  return wgrad_impl(x, dy, dw, ws, nullptr, B, C, H, W, M, in_mode, opts, stream, phase);
}
// The same with V = B^T x B supplied by the caller (the first 36 * C * P floats of the forward call's workspace, kept alive):
// the input transform is skipped.
extern "C" int prn_conv3x3_winograd_wgrad_v(const float* V, const float* dy, float* dw, void* ws, int B, int C, int H, int W, int M, const prn_gemm_opts* opts,
                                            void* stream) {
  PRN_REQUIRE(V, "prn_conv3x3_winograd_wgrad_v: null V");
  return wgrad_impl(nullptr, dy, dw, ws, nullptr, B, C, H, W, M, PRN_IN_ZERO, opts, stream, 0, V);
}
extern "C" int prn_conv3x3_winograd_wgrad_ragged(const float* x, const float* dy, float* dw, void* ws, const prn_ragged* rg, int B, int C, int M,
                                                 const prn_gemm_opts* opts, void* stream) {
  PRN_REQUIRE(rg, "prn_conv3x3_winograd_wgrad_ragged: null segment table");
  return wgrad_impl(x, dy, dw, ws, rg, B, C, 0, 0, M, PRN_IN_ZERO, opts, stream, 0);
}

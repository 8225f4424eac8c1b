// BatchNorm2d (+residual +ReLU) and GroupNorm(32)+ReLU, forward and backward. HBM-bound: every pass is a
// float4-vectorised stream over NCHW planes with wave-level (shuffle) reductions; cross-block combines are
// done in fp64 from fixed-order partials, so results are deterministic.
#include <stdlib.h>
#include "prn_common.h"

namespace {


// ------------------------------------------------------------------------------------------- BatchNorm
// grid (C, S): block (c, s) reduces slice s of every image plane of channel c.  S is chosen by the launcher so that a
// block owns >= ~8k elements (one block per channel for the 15x20 maps, up to PRN_BN_SPLITS for 120x160 ones);
// float4 streams whenever HW % 4 == 0.
__device__ __forceinline__ void slice_bounds(int HW, int s, int S, bool vec, int& beg, int& end) {
  if (vec) {
    const int n4 = HW >> 2;
    beg = (int)((int64_t)n4 * s / S) * 4;
    end = (int)((int64_t)n4 * (s + 1) / S) * 4;
  } else {
    beg = (int)((int64_t)HW * s / S);
    end = (int)((int64_t)HW * (s + 1) / S);
  }
}

__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, double* __restrict__ ws, int B, int C, int HW) {
  const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
  const bool vec = (HW & 3) == 0;
  int beg, end;
  slice_bounds(HW, s, S, vec, beg, end);
  float s1 = 0.f, s2 = 0.f;
  for (int b = 0; b < B; ++b) {
    const float* xp = x + ((size_t)b * C + c) * HW;
    if (vec) {
      for (int p = beg + threadIdx.x * 4; p < end; p += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(xp + p);
        s1 += (v.x + v.y) + (v.z + v.w);
        s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
    } else {
      for (int p = beg + threadIdx.x; p < end; p += 256) {
        const float v = xp[p];
        s1 += v; s2 += v * v;
      }
    }
  }
  __shared__ double sm[16];
  double d1 = s1, d2 = s2;
  block_sum2(d1, d2, sm);
  if (threadIdx.x == 0) { ws[((size_t)c * S + s) * 2] = d1; ws[((size_t)c * S + s) * 2 + 1] = d2; }
}

__global__ void bn_finalize_kernel(const double* __restrict__ ws, float* __restrict__ stats, float* __restrict__ rmean,
                                   float* __restrict__ rvar, int C, int S, double count, float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int s = 0; s < S; ++s) { s1 += ws[((size_t)c * S + s) * 2]; s2 += ws[((size_t)c * S + s) * 2 + 1]; }
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[c] = (float)mean;
  stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
  if (rmean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
  }
}

// y = relu?(x*scale + shift + res?) ; one block row per (b,c) plane chunk, float4 when HW % 4 == 0
// FIN: training forward -- every block first reduces its channel's S fixed-order fp64 partials (written by bn_partial_kernel)
// to mean / invstd itself, so no separate finalize launch is needed; block (0, image 0) of each channel publishes the
// statistics and updates the running buffers.
template <bool FIN>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ res, float* __restrict__ y, int C, int HW, int relu,
                                                       const double* __restrict__ ws, float* __restrict__ rmean, float* __restrict__ rvar,
                                                       int S, double count, float eps, float momentum, int64_t ybs) {
  const int bc = blockIdx.y, c = bc % C;
  float mean_f, istd_f;
  if (FIN) {
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < S; ++s) { s1 += ws[((size_t)c * S + s) * 2]; s2 += ws[((size_t)c * S + s) * 2 + 1]; }
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_f = (float)mean;
    istd_f = (float)(1.0 / sqrt(var + (double)eps));
    if (blockIdx.x == 0 && bc == c && threadIdx.x == 0) {
      stats[c] = mean_f;
      stats[C + c] = istd_f;
      if (rmean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean_f;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
      }
    }
  } else {
    mean_f = stats[c];
    istd_f = stats[C + c];
  }
  const float sc = istd_f * gamma[c], sh = beta[c] - mean_f * sc;
  const size_t base = (size_t)bc * HW;
  y += (size_t)(bc / C) * ybs + (size_t)c * HW - base;      // y may be a channel slice of a wider tensor (batch stride ybs)
  if ((HW & 3) == 0) {
    const int n4 = HW >> 2;
    const float4* xp = reinterpret_cast<const float4*>(x + base);
    const float4* rp = res ? reinterpret_cast<const float4*>(res + base) : nullptr;
    float4* yp = reinterpret_cast<float4*>(y + base);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
      float4 v = xp[i];
      v.x = fmaf(v.x, sc, sh); v.y = fmaf(v.y, sc, sh); v.z = fmaf(v.z, sc, sh); v.w = fmaf(v.w, sc, sh);
      if (rp) { const float4 r = rp[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      yp[i] = v;
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
      float v = fmaf(x[base + i], sc, sh);
      if (res) v += res[base + i];
      if (relu) v = fmaxf(v, 0.f);
      y[base + i] = v;
    }
  }
}

// partial sums of g and g*xhat, g = dy * (y > 0 if relu)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ y, const float* __restrict__ stats,
                                                             double* __restrict__ ws, int B, int C, int HW, int relu, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int64_t dbs) {
  // relu: 0 none | 1 mask = (y > 0) read from the forward output | 2 mask recomputed as fmaf(x, sc, sh) > 0 (the forward's
  // own expression; BN + ReLU without a residual), which saves the y read in both backward passes
  const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
  const bool vec = (HW & 3) == 0;
  int beg, end;
  slice_bounds(HW, s, S, vec, beg, end);
  const float mean = stats[c], istd = stats[C + c];
  const float sc = relu == 2 ? istd * gamma[c] : 0.f, sh = relu == 2 ? beta[c] - mean * sc : 0.f;
  float s1 = 0.f, s2 = 0.f;
  for (int b = 0; b < B; ++b) {
    const size_t base = ((size_t)b * C + c) * HW;
    const float* dyb = dy + (size_t)b * dbs + (size_t)c * HW;      // dy may be a channel slice of a wider tensor
    if (vec) {
      for (int p = beg + threadIdx.x * 4; p < end; p += 1024) {
        float4 g = *reinterpret_cast<const float4*>(dyb + p);
        const float4 xv = *reinterpret_cast<const float4*>(x + base + p);
        if (relu == 1) {
          const float4 yv = *reinterpret_cast<const float4*>(y + base + p);
          g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
        } else if (relu == 2) {
          g.x = fmaf(xv.x, sc, sh) > 0.f ? g.x : 0.f; g.y = fmaf(xv.y, sc, sh) > 0.f ? g.y : 0.f;
          g.z = fmaf(xv.z, sc, sh) > 0.f ? g.z : 0.f; g.w = fmaf(xv.w, sc, sh) > 0.f ? g.w : 0.f;
        }
        s1 += (g.x + g.y) + (g.z + g.w);
        s2 += (g.x * (xv.x - mean) + g.y * (xv.y - mean)) + (g.z * (xv.z - mean) + g.w * (xv.w - mean));
      }
    } else {
      for (int p = beg + threadIdx.x; p < end; p += 256) {
        float g = dyb[p];
        if (relu == 1 && !(y[base + p] > 0.f)) g = 0.f;
        if (relu == 2 && !(fmaf(x[base + p], sc, sh) > 0.f)) g = 0.f;
        s1 += g; s2 += g * (x[base + p] - mean);
      }
    }
  }
  s2 *= istd;
  __shared__ double sm[16];
  double d1 = s1, d2 = s2;
  block_sum2(d1, d2, sm);
  if (threadIdx.x == 0) { ws[((size_t)c * S + s) * 2] = d1; ws[((size_t)c * S + s) * 2 + 1] = d2; }
}

// dx = gamma*istd*(g - sum_g/N - xhat*sum_gx/N)   (training)   |   dx = gamma*istd*g   (frozen / eval stats)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ y, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, const double* __restrict__ ws,
                                                           float* __restrict__ dx, float* __restrict__ dres, int C, int HW,
                                                           int S, float inv_count, int relu, int frozen, int have_partials,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           const float* __restrict__ beta, int64_t dbs) {
  const int bc = blockIdx.y, c = bc % C;
  const float mean = stats[c], istd = stats[C + c], gi = gamma[c] * istd;
  const float sc = relu == 2 ? gi : 0.f, sh = relu == 2 ? beta[c] - mean * sc : 0.f;     // (see bn_bwd_partial_kernel)
  double t1 = 0.0, t2 = 0.0;                      // every block sums its channel's fixed-order partials (no finalize launch)
  if (have_partials)
    for (int s = 0; s < S; ++s) { t1 += ws[((size_t)c * S + s) * 2]; t2 += ws[((size_t)c * S + s) * 2 + 1]; }
  if (have_partials && blockIdx.x == 0 && bc == c && threadIdx.x == 0) {
    if (dbeta) dbeta[c] = (float)t1;
    if (dgamma) dgamma[c] = (float)t2;
  }
  const float m1 = frozen ? 0.f : (float)t1 * inv_count;
  const float m2 = frozen ? 0.f : (float)t2 * inv_count;
  const size_t base = (size_t)bc * HW;
  dy += (size_t)(bc / C) * dbs + (size_t)c * HW - base;
  if ((HW & 3) == 0) {
    const int n4 = HW >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
      float4 g = reinterpret_cast<const float4*>(dy + base)[i];
      const float4 xv = reinterpret_cast<const float4*>(x + base)[i];
      if (relu == 1) {
        const float4 yv = reinterpret_cast<const float4*>(y + base)[i];
        g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
      } else if (relu == 2) {
        g.x = fmaf(xv.x, sc, sh) > 0.f ? g.x : 0.f; g.y = fmaf(xv.y, sc, sh) > 0.f ? g.y : 0.f;
        g.z = fmaf(xv.z, sc, sh) > 0.f ? g.z : 0.f; g.w = fmaf(xv.w, sc, sh) > 0.f ? g.w : 0.f;
      }
      if (dres) reinterpret_cast<float4*>(dres + base)[i] = g;
      float4 o;
      o.x = gi * (g.x - m1 - (xv.x - mean) * istd * m2); o.y = gi * (g.y - m1 - (xv.y - mean) * istd * m2);
      o.z = gi * (g.z - m1 - (xv.z - mean) * istd * m2); o.w = gi * (g.w - m1 - (xv.w - mean) * istd * m2);
      reinterpret_cast<float4*>(dx + base)[i] = o;
    }
    return;
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    float g = dy[base + i];
    if (relu == 1 && !(y[base + i] > 0.f)) g = 0.f;
    if (relu == 2 && !(fmaf(x[base + i], sc, sh) > 0.f)) g = 0.f;
    if (dres) dres[base + i] = g;
    const float xh = (x[base + i] - mean) * istd;
    dx[base + i] = gi * (g - m1 - xh * m2);
  }
}

// ------------------------------------------------------------------------------------------- BatchNorm, one pass
// Small maps (stages 3-4 of the backbone: 8 x 30x40 / 15x20 pixels per channel): one workgroup owns a whole channel and
// keeps its <= 12288 values in REGISTERS between the statistics and the normalisation, so the activation is read from HBM
// once and the layer is one launch instead of two -- for 78 of the 113 BatchNorm layers of PlaneRecNet_101, whose two
// launches were ~10 us each, i.e. launch-bound.  NV = float4 groups per thread; needs HW % 4 == 0.
constexpr int BN_SMALL_MAX = 12288;

// The Winograd INPUT transform of a channel whose plane (all B images, <= BN_SMALL_MAX floats) a one-launch BatchNorm kernel has just put in LDS: what
// winograd_input_kernel<PRN_IN_ZERO> (prn_winograd.hip) computes from global memory -- V[(i*6+j)][c][p] = (B^T d B)[i][j] of the 6x6 patch of tile p -- for
// the 3x3 / pad-1 convolution that consumes the kernel's output (models/backbone.py:57-60: bn1 -> conv2; in the backward pass bn2's input gradient ->
// conv2's input-gradient convolution).  Dense batch, W % 4 == 0.
__device__ __forceinline__ void plane_to_winograd_v(const float* __restrict__ plane, float* __restrict__ V, int c, int C, int B, int H, int W) {
  const int TW = W >> 2, TH = (H + 3) >> 2, P = B * TH * TW, P4 = (P + 3) & ~3, HW = H * W;
  const size_t zs = (size_t)C * P4;
  for (int p = threadIdx.x; p < P; p += 256) {
    const int tx = p % TW, ty = (p / TW) % TH, b = p / (TW * TH);
    const float* pb = plane + b * HW;
    const int w0 = 4 * tx;
    float d[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int ih = 4 * ty - 1 + i;
      const bool ok = (unsigned)ih < (unsigned)H;
      const float* row = pb + (ok ? ih : 0) * W;
      const float4 q = *reinterpret_cast<const float4*>(row + w0);
      const bool okl = w0 > 0, okr = w0 + 4 < W;
      const float l = row[okl ? w0 - 1 : 0], r = row[okr ? w0 + 4 : 0];
      d[i][0] = (ok && okl) ? l : 0.f;
      d[i][1] = ok ? q.x : 0.f; d[i][2] = ok ? q.y : 0.f; d[i][3] = ok ? q.z : 0.f; d[i][4] = ok ? q.w : 0.f;
      d[i][5] = (ok && okr) ? r : 0.f;
    }
    float t[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float v[6], o[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) v[i] = d[i][j];
      bt6(v, o);
#pragma unroll
      for (int i = 0; i < 6; ++i) t[i][j] = o[i];
    }
    float* out = V + (size_t)c * P4 + p;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float o[6];
      bt6(t[i], o);
#pragma unroll
      for (int j = 0; j < 6; ++j) out[(size_t)(i * 6 + j) * zs] = o[j];
    }
  }
}

template <int NV, bool WINO = false>
__global__ __launch_bounds__(256) void bn_small_fwd_kernel(const float* __restrict__ x, float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ res, float* __restrict__ y,
                                                           float* __restrict__ rmean, float* __restrict__ rvar, int B, int C, int HW, float eps,
                                                           float momentum, int relu, int64_t ybs, int nparts, int64_t pstride, float* __restrict__ xsum,
                                                           float* __restrict__ V = nullptr, int H = 0, int W = 0) {
  // nparts > 1: x is the first of `nparts` K-split partial sums of the producing GEMM (pstride elements apart); they are summed here in split
  // order -- bit for bit what reduce_epilogue_kernel would have written -- and the sum goes to xsum (the BatchNorm input the backward reads).
  __shared__ float wino_plane[WINO ? BN_SMALL_MAX : 4];
  const int c = blockIdx.x, n4 = (B * HW) >> 2;
  const int64_t ydelta = ybs - (int64_t)C * HW;             // y as a channel slice of a wider tensor: extra elements per image
  // All NV loads are issued before anything is consumed: they are UNCONDITIONAL (a lane past the end re-reads element 0 and
  // zeroes it afterwards).  With "ok ? load : 0" hipcc wrapped every load in an exec-masked branch and waited for it
  // (s_waitcnt vmcnt(0)) before the next one: NV dependent memory round trips per thread (bn_small_bwd: 19.7 us per launch).
  float4 v[NV];
  size_t off[NV];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int q = threadIdx.x + i * 256;
    const int e = q < n4 ? q * 4 : 0, b = e / HW, p = e - b * HW;
    off[i] = ((size_t)b * C + c) * HW + p;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(x + off[i]);
  __builtin_amdgcn_sched_barrier(0);
  for (int s = 1; s < nparts; ++s) {
    float4 t[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) t[i] = *reinterpret_cast<const float4*>(x + (size_t)s * pstride + off[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i].x += t[i].x; v[i].y += t[i].y; v[i].z += t[i].z; v[i].w += t[i].w; }
  }
  if (xsum) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (threadIdx.x + i * 256 < n4) *reinterpret_cast<float4*>(xsum + off[i]) = v[i];
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (threadIdx.x + i * 256 >= n4) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    s1 += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    s2 += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  __shared__ double sm[16];
  double d1 = s1, d2 = s2;
  block_sum2(d1, d2, sm);
  const double count = (double)B * HW, mean = d1 / count;
  double var = d2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float mean_f = (float)mean, istd_f = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0) {
    stats[c] = mean_f;
    stats[C + c] = istd_f;
    if (rmean) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean_f;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
  }
  const float sc = istd_f * gamma[c], sh = beta[c] - mean_f * sc;
  float4 r[NV];
  if (res) {                                               // (uniform) residual: again all loads first
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = *reinterpret_cast<const float4*>(res + off[i]);
    __builtin_amdgcn_sched_barrier(0);
  } else {
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (threadIdx.x + i * 256 >= n4) continue;
    float4 o = v[i];
    o.x = fmaf(o.x, sc, sh) + r[i].x; o.y = fmaf(o.y, sc, sh) + r[i].y; o.z = fmaf(o.z, sc, sh) + r[i].z; o.w = fmaf(o.w, sc, sh) + r[i].w;
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    size_t oy = off[i];
    if (ydelta) oy += (size_t)(((threadIdx.x + i * 256) * 4) / HW) * ydelta;
    *reinterpret_cast<float4*>(y + oy) = o;
    if constexpr (WINO) *reinterpret_cast<float4*>(&wino_plane[(threadIdx.x + i * 256) * 4]) = o;
  }
  if constexpr (WINO) {                                      // WINO: the output also leaves as B^T y B for the 3x3 convolution behind this layer
    __syncthreads();
    plane_to_winograd_v(wino_plane, V, c, C, B, H, W);
  }
}

// relu: 0 none | 1 mask from y | 2 mask from fmaf(x, sc, sh) (no residual)
template <int NV, bool WINO = false>
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ dx, float* __restrict__ dres,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C, int HW, int relu,
                                                           int frozen, int64_t dbs, int nparts, int64_t pstride, float* __restrict__ V = nullptr, int H = 0,
                                                           int W = 0) {
  // nparts > 1: dy is the first of `nparts` dense K-split partial sums of the input-gradient GEMM that produced it (see bn_small_fwd_kernel)
  __shared__ float wino_plane[WINO ? BN_SMALL_MAX : 4];
  const int c = blockIdx.x, n4 = (B * HW) >> 2;
  const int64_t ddelta = dbs - (int64_t)C * HW;
  const float mean = stats[c], istd = stats[C + c], gi = gamma[c] * istd;
  const float sc = relu == 2 ? gi : 0.f, sh = relu == 2 ? beta[c] - mean * sc : 0.f;
  float4 g[NV], xv[NV];
  size_t off[NV];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int q = threadIdx.x + i * 256;
    const int e = q < n4 ? q * 4 : 0, b = e / HW, p = e - b * HW;
    off[i] = ((size_t)b * C + c) * HW + p;
  }
  // unconditional loads, all issued before use (see bn_small_fwd_kernel)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int q = threadIdx.x + i * 256;
    g[i] = *reinterpret_cast<const float4*>(dy + off[i] + (ddelta ? (size_t)((q < n4 ? q * 4 : 0) / HW) * ddelta : 0));
    xv[i] = *reinterpret_cast<const float4*>(x + off[i]);
  }
  __builtin_amdgcn_sched_barrier(0);
  for (int s = 1; s < nparts; ++s) {
    float4 t[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) t[i] = *reinterpret_cast<const float4*>(dy + (size_t)s * pstride + off[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NV; ++i) { g[i].x += t[i].x; g[i].y += t[i].y; g[i].z += t[i].z; g[i].w += t[i].w; }
  }
  if (relu == 1) {                                         // mask from the forward output (a residual was added)
    float4 yv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) yv[i] = *reinterpret_cast<const float4*>(y + off[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      g[i].x = yv[i].x > 0.f ? g[i].x : 0.f; g[i].y = yv[i].y > 0.f ? g[i].y : 0.f;
      g[i].z = yv[i].z > 0.f ? g[i].z : 0.f; g[i].w = yv[i].w > 0.f ? g[i].w : 0.f;
    }
  } else if (relu == 2) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      g[i].x = fmaf(xv[i].x, sc, sh) > 0.f ? g[i].x : 0.f; g[i].y = fmaf(xv[i].y, sc, sh) > 0.f ? g[i].y : 0.f;
      g[i].z = fmaf(xv[i].z, sc, sh) > 0.f ? g[i].z : 0.f; g[i].w = fmaf(xv[i].w, sc, sh) > 0.f ? g[i].w : 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (threadIdx.x + i * 256 >= n4) { g[i] = make_float4(0.f, 0.f, 0.f, 0.f); xv[i] = g[i]; }
    s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
    s2 += (g[i].x * (xv[i].x - mean) + g[i].y * (xv[i].y - mean)) + (g[i].z * (xv[i].z - mean) + g[i].w * (xv[i].w - mean));
  }
  s2 *= istd;
  __shared__ double sm[16];
  double t1 = s1, t2 = s2;
  block_sum2(t1, t2, sm);
  if (threadIdx.x == 0) {
    if (dbeta) dbeta[c] = (float)t1;
    if (dgamma) dgamma[c] = (float)t2;
  }
  const float inv_count = 1.f / ((float)B * HW);
  const float m1 = frozen ? 0.f : (float)t1 * inv_count, m2 = frozen ? 0.f : (float)t2 * inv_count;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (threadIdx.x + i * 256 >= n4) continue;
    if (dres) *reinterpret_cast<float4*>(dres + off[i]) = g[i];
    float4 o;
    o.x = gi * (g[i].x - m1 - (xv[i].x - mean) * istd * m2); o.y = gi * (g[i].y - m1 - (xv[i].y - mean) * istd * m2);
    o.z = gi * (g[i].z - m1 - (xv[i].z - mean) * istd * m2); o.w = gi * (g[i].w - m1 - (xv[i].w - mean) * istd * m2);
    *reinterpret_cast<float4*>(dx + off[i]) = o;
    if constexpr (WINO) *reinterpret_cast<float4*>(&wino_plane[(threadIdx.x + i * 256) * 4]) = o;
  }
  if constexpr (WINO) {                                      // WINO: dx also leaves as B^T dx B for the input-gradient convolution of the 3x3 layer in front
    __syncthreads();
    plane_to_winograd_v(wino_plane, V, c, C, B, H, W);
  }
}

bool bn_small_ok(int B, int HW) {
  static const int on = prn_env_int("PRN_BN_SMALL", 1);                                   // PRN_BN_SMALL=0 switches the one-pass kernels off (A/B)
  return on && (HW & 3) == 0 && (int64_t)B * HW <= BN_SMALL_MAX;
}
}  // namespace

// 1: the training-mode forward / backward of a [B, C, HW] BatchNorm run as ONE launch that reads the activation once (the channel
// lives in a workgroup's registers); 0: two launches (statistics pass + apply pass) -- for profilers that credit executed bytes.
extern "C" int prn_bn_kernel_kind(int B, int HW) { return bn_small_ok(B, HW) ? 1 : 0; }

namespace {

// ------------------------------------------------------------------------------------------- GroupNorm
// one block per (b, group): pass 1 statistics, pass 2 normalise + affine + ReLU (second read is L2-resident)
// Ragged batch (nseg > 0): segment s is a dense [B, C, hw[s]] tensor at element offset off[s]; block -> (segment, image, group)
struct GnSeg { int nseg; int hw[6]; int off[6]; };
__device__ __forceinline__ int gn_seg_get(const int (&v)[6], int s) {
  int r = v[0];
#pragma unroll
  for (int t = 1; t < 6; ++t)
    if (s == t) r = v[t];
  return r;
}

__global__ __launch_bounds__(1024) void gn_relu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y,
                                                          float* __restrict__ stats, int C, int HW, int G, float eps, int BG, GnSeg sg) {
  int bg = blockIdx.x;
  if (sg.nseg > 0) {
    const int sI = bg / BG;
    bg -= sI * BG;
    HW = gn_seg_get(sg.hw, sI);
    const int off = gn_seg_get(sg.off, sI);
    x += off; y += off; stats += (size_t)sI * BG * 2;
  }
  const int g = bg % G, cpg = C / G;
  const int b = bg / G;
  const size_t base = ((size_t)b * C + (size_t)g * cpg) * HW;
  const int n = cpg * HW;
  const bool vec = (HW & 3) == 0 && (base & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  float s1 = 0.f, s2 = 0.f;
  if (vec) {
    for (int i = threadIdx.x * 4; i < n; i += blockDim.x * 4) {
      const float4 v = *reinterpret_cast<const float4*>(x + base + i);
      s1 += (v.x + v.y) + (v.z + v.w);
      s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
  } else {
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const float v = x[base + i]; s1 += v; s2 += v * v; }
  }
  __shared__ double sm[32];
  double d1 = s1, d2 = s2;
  block_sum2(d1, d2, sm);
  const double mean = d1 / n;
  double var = d2 / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float mu = (float)mean, istd = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0) { stats[bg * 2] = mu; stats[bg * 2 + 1] = istd; }
  if (vec) {
    for (int i = threadIdx.x * 4; i < n; i += blockDim.x * 4) {
      const int c = g * cpg + i / HW;
      const float gc = gamma[c], bc = beta[c];
      const float4 v = *reinterpret_cast<const float4*>(x + base + i);
      float4 o;
      o.x = fmaxf(((v.x - mu) * istd) * gc + bc, 0.f); o.y = fmaxf(((v.y - mu) * istd) * gc + bc, 0.f);
      o.z = fmaxf(((v.z - mu) * istd) * gc + bc, 0.f); o.w = fmaxf(((v.w - mu) * istd) * gc + bc, 0.f);
      *reinterpret_cast<float4*>(y + base + i) = o;
    }
    return;
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int c = g * cpg + i / HW;
    const float xh = (x[base + i] - mu) * istd;
    const float v = xh * gamma[c] + beta[c];               // (the backward re-derives the ReLU mask from this same expression)
    y[base + i] = fmaxf(v, 0.f);
  }
}

// The ReLU mask is re-derived from x (sign of the forward's own (x - mu) * istd * gamma + beta): the forward output is not
// read at all (4 instead of 6 activation reads over the two passes).
__global__ __launch_bounds__(1024) void gn_relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ beta, const float* __restrict__ stats,
                                                          const float* __restrict__ gamma, float* __restrict__ dx,
                                                          float* __restrict__ dgp, float* __restrict__ dbp, int C, int HW, int G, int BG,
                                                          GnSeg sg) {
  int bg = blockIdx.x;
  if (sg.nseg > 0) {
    const int sI = bg / BG;
    bg -= sI * BG;
    HW = gn_seg_get(sg.hw, sI);
    const int off = gn_seg_get(sg.off, sI);
    dy += off; x += off; dx += off; stats += (size_t)sI * BG * 2;
    dgp += (size_t)sI * (BG / G) * C; dbp += (size_t)sI * (BG / G) * C;
  }
  const int g = bg % G, cpg = C / G, b = bg / G;
  const size_t base = ((size_t)b * C + (size_t)g * cpg) * HW;
  const float mu = stats[bg * 2], istd = stats[bg * 2 + 1];
  __shared__ double sm[32];
  __shared__ double tot[2];
  const bool vec = (HW & 3) == 0 && (base & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
  double ds = 0.0, db = 0.0;       // sum_c gamma_c * sum(g*xhat), sum_c gamma_c * sum(g)
  for (int cc = 0; cc < cpg; ++cc) {
    const int c = g * cpg + cc;
    const size_t cb = base + (size_t)cc * HW;
    float s1 = 0.f, s2 = 0.f;
    const float gc = gamma[c], bc = beta[c];
    if (vec) {
      for (int i = threadIdx.x * 4; i < HW; i += blockDim.x * 4) {
        float4 gg = *reinterpret_cast<const float4*>(dy + cb + i);
        const float4 xv = *reinterpret_cast<const float4*>(x + cb + i);
        const float h0 = (xv.x - mu) * istd, h1 = (xv.y - mu) * istd, h2 = (xv.z - mu) * istd, h3 = (xv.w - mu) * istd;
        if (!(h0 * gc + bc > 0.f)) gg.x = 0.f;
        if (!(h1 * gc + bc > 0.f)) gg.y = 0.f;
        if (!(h2 * gc + bc > 0.f)) gg.z = 0.f;
        if (!(h3 * gc + bc > 0.f)) gg.w = 0.f;
        s1 += (gg.x + gg.y) + (gg.z + gg.w);
        s2 += (gg.x * h0 + gg.y * h1) + (gg.z * h2 + gg.w * h3);
      }
    } else {
      for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        float gg = dy[cb + i];
        const float xh = (x[cb + i] - mu) * istd;
        if (!(xh * gc + bc > 0.f)) gg = 0.f;
        s1 += gg; s2 += gg * xh;
      }
    }
    double d1 = s1, d2 = s2;
    block_sum2(d1, d2, sm);
    if (threadIdx.x == 0) { dbp[(size_t)b * C + c] = (float)d1; dgp[(size_t)b * C + c] = (float)d2; }
    ds += (double)gamma[c] * d2;
    db += (double)gamma[c] * d1;
  }
  const int n = cpg * HW;
  const float m1 = (float)(db / n), m2 = (float)(ds / n);
  if (vec) {
    for (int i = threadIdx.x * 4; i < n; i += blockDim.x * 4) {
      const int c = g * cpg + i / HW;
      const float gc = gamma[c], bc = beta[c];
      float4 gg = *reinterpret_cast<const float4*>(dy + base + i);
      const float4 xv = *reinterpret_cast<const float4*>(x + base + i);
      const float h0 = (xv.x - mu) * istd, h1 = (xv.y - mu) * istd, h2 = (xv.z - mu) * istd, h3 = (xv.w - mu) * istd;
      if (!(h0 * gc + bc > 0.f)) gg.x = 0.f;
      if (!(h1 * gc + bc > 0.f)) gg.y = 0.f;
      if (!(h2 * gc + bc > 0.f)) gg.z = 0.f;
      if (!(h3 * gc + bc > 0.f)) gg.w = 0.f;
      float4 o;
      o.x = istd * (gc * gg.x - m1 - h0 * m2); o.y = istd * (gc * gg.y - m1 - h1 * m2);
      o.z = istd * (gc * gg.z - m1 - h2 * m2); o.w = istd * (gc * gg.w - m1 - h3 * m2);
      *reinterpret_cast<float4*>(dx + base + i) = o;
    }
    return;
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int c = g * cpg + i / HW;
    float gg = dy[base + i];
    const float xh = (x[base + i] - mu) * istd;
    if (!(xh * gamma[c] + beta[c] > 0.f)) gg = 0.f;
    dx[base + i] = istd * (gamma[c] * gg - m1 - xh * m2);
  }
  (void)tot;
}

// splits per channel: >= ~8k elements per block, at most PRN_BN_SPLITS (the workspace contract)
int bn_splits(int B, int HW) {
  int s = (int)(((int64_t)B * HW + 8191) / 8192);
  if (s > PRN_BN_SPLITS) s = PRN_BN_SPLITS;
  if (s < 1) s = 1;
  if (s > HW / 4 && HW >= 4) s = HW / 4;
  return s < 1 ? 1 : s;
}

}  // namespace

extern "C" int prn_bn_stats(const float* x, float* stats, float* running_mean, float* running_var, double* ws,
                            int B, int C, int HW, float eps, float momentum, void* stream) {
  PRN_REQUIRE(x && stats && ws && B > 0 && C > 0 && HW > 0, "prn_bn_stats: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int S = bn_splits(B, HW);
  hipLaunchKernelGGL(bn_partial_kernel, dim3(C, S), dim3(256), 0, st, x, ws, B, C, HW);
  PRN_CHECK_LAUNCH("prn_bn_stats/partial");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, (const double*)ws, stats, running_mean, running_var, C,
                     S, (double)B * HW, eps, momentum);
  PRN_CHECK_LAUNCH("prn_bn_stats/finalize");
  return 0;
}

extern "C" int prn_bn_apply(const float* x, const float* stats, const float* gamma, const float* beta, const float* residual,
                            float* y, int B, int C, int HW, int relu, void* stream) {
  PRN_REQUIRE(x && stats && gamma && beta && y && B > 0 && C > 0 && HW > 0, "prn_bn_apply: bad arguments");
  int gx = cdiv(HW, 256 * 8);
  if (gx < 1) gx = 1;
  PRN_REQUIRE((int64_t)B * C <= 65535, "prn_bn_apply: B*C too large for grid.y");
  hipLaunchKernelGGL((bn_apply_kernel<false>), dim3(gx, B * C), dim3(256), 0, (hipStream_t)stream, x, const_cast<float*>(stats), gamma, beta,
                     residual, y, C, HW, relu, (const double*)nullptr, (float*)nullptr, (float*)nullptr, 0, 0.0, 0.f, 0.f, (int64_t)C * HW);
  PRN_CHECK_LAUNCH("prn_bn_apply");
  return 0;
}

extern "C" int prn_bn_train_fwd(const float* x, float* stats, const float* gamma, const float* beta, const float* residual, float* y,
                                float* running_mean, float* running_var, double* ws, int B, int C, int HW, float eps, float momentum,
                                int relu, void* stream) {
  return prn_bn_train_fwd_into(x, stats, gamma, beta, residual, y, (int64_t)C * HW, running_mean, running_var, ws, B, C, HW, eps, momentum, relu, stream);
}

extern "C" int prn_bn_train_fwd_into(const float* x, float* stats, const float* gamma, const float* beta, const float* residual, float* y,
                                     int64_t y_batch_stride, float* running_mean, float* running_var, double* ws, int B, int C, int HW, float eps,
                                     float momentum, int relu, void* stream) {
  PRN_REQUIRE(x && stats && gamma && beta && y && B > 0 && C > 0 && HW > 0, "prn_bn_train_fwd: bad arguments");
  PRN_REQUIRE(ws || bn_small_ok(B, HW), "prn_bn_train_fwd: the two-launch path needs its workspace (prn_bn_kernel_kind(B, HW) == 0)");
  PRN_REQUIRE(y_batch_stride >= (int64_t)C * HW && ((HW & 3) || ((y_batch_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0)),
              "prn_bn_train_fwd_into: bad output batch stride / alignment");
  const int64_t ybs = y_batch_stride;
  PRN_REQUIRE((int64_t)B * C <= 65535, "prn_bn_train_fwd: B*C too large for grid.y");
  hipStream_t st = (hipStream_t)stream;
  if (bn_small_ok(B, HW)) {                                // whole channel in one workgroup's registers: one launch, one read
    const int nv = cdiv(B * HW / 4, 256);
#define PRN_BN_SMALL_FWD(NV_) hipLaunchKernelGGL((bn_small_fwd_kernel<NV_>), dim3(C), dim3(256), 0, st, x, stats, gamma, beta, residual, y, \
                                                 running_mean, running_var, B, C, HW, eps, momentum, relu, ybs, 1, (int64_t)0, (float*)nullptr)
    if (nv <= 3) PRN_BN_SMALL_FWD(3); else if (nv <= 6) PRN_BN_SMALL_FWD(6); else if (nv <= 10) PRN_BN_SMALL_FWD(10); else PRN_BN_SMALL_FWD(12);
#undef PRN_BN_SMALL_FWD
    PRN_CHECK_LAUNCH("prn_bn_train_fwd/small");
    return 0;
  }
  const int S = bn_splits(B, HW);
  for (int r = PRN_REPS(1); r > 0; --r) hipLaunchKernelGGL(bn_partial_kernel, dim3(C, S), dim3(256), 0, st, x, ws, B, C, HW);
  PRN_CHECK_LAUNCH("prn_bn_train_fwd/partial");
  int gx = cdiv(HW, 256 * 8);
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL((bn_apply_kernel<true>), dim3(gx, B * C), dim3(256), 0, st, x, stats, gamma, beta, residual, y, C, HW, relu,
                     (const double*)ws, running_mean, running_var, S, (double)B * HW, eps, momentum, ybs);
  PRN_CHECK_LAUNCH("prn_bn_train_fwd/apply");
  return 0;
}

extern "C" int prn_bn_bwd(const float* dy, const float* x, const float* y, const float* stats, const float* gamma, const float* beta,
                          float* dx, float* dres, float* dgamma, float* dbeta, double* ws,
                          int B, int C, int HW, int relu, int frozen, void* stream) {
  return prn_bn_bwd_from(dy, (int64_t)C * HW, x, y, stats, gamma, beta, dx, dres, dgamma, dbeta, ws, B, C, HW, relu, frozen, stream);
}

extern "C" int prn_bn_bwd_from(const float* dy, int64_t dy_batch_stride, const float* x, const float* y, const float* stats, const float* gamma,
                               const float* beta, float* dx, float* dres, float* dgamma, float* dbeta, double* ws,
                               int B, int C, int HW, int relu, int frozen, void* stream) {
  PRN_REQUIRE(dy && x && stats && gamma && dx && B > 0 && C > 0 && HW > 0, "prn_bn_bwd: bad arguments");
  PRN_REQUIRE(ws || bn_small_ok(B, HW), "prn_bn_bwd: the two-launch path needs its workspace (prn_bn_kernel_kind(B, HW) == 0)");
  PRN_REQUIRE(dy_batch_stride >= (int64_t)C * HW && ((HW & 3) || ((dy_batch_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0)),
              "prn_bn_bwd_from: bad gradient batch stride / alignment");
  const int64_t dbs = dy_batch_stride;
  PRN_REQUIRE(!relu || y || (beta && !dres), "prn_bn_bwd: relu needs the forward output, or beta (and no residual) to recompute its sign");
  if (relu) relu = y ? 1 : 2;
  PRN_REQUIRE((int64_t)B * C <= 65535, "prn_bn_bwd: B*C too large for grid.y");
  hipStream_t st = (hipStream_t)stream;
  if (bn_small_ok(B, HW)) {
    const int nv = cdiv(B * HW / 4, 256);
#define PRN_BN_SMALL_BWD(NV_) hipLaunchKernelGGL((bn_small_bwd_kernel<NV_>), dim3(C), dim3(256), 0, st, dy, x, y, stats, gamma, beta, dx, dres, dgamma, \
                                                 dbeta, B, C, HW, relu, frozen, dbs, 1, (int64_t)0)
    if (nv <= 3) PRN_BN_SMALL_BWD(3); else if (nv <= 6) PRN_BN_SMALL_BWD(6); else if (nv <= 10) PRN_BN_SMALL_BWD(10); else PRN_BN_SMALL_BWD(12);
#undef PRN_BN_SMALL_BWD
    PRN_CHECK_LAUNCH("prn_bn_bwd/small");
    return 0;
  }
  const int S = bn_splits(B, HW);
  const int have = (!frozen || dgamma || dbeta) ? 1 : 0;
  if (have) {
    for (int r = PRN_REPS(2); r > 0; --r) hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(C, S), dim3(256), 0, st, dy, x, y, stats, ws, B, C, HW, relu, gamma, beta, dbs);
    PRN_CHECK_LAUNCH("prn_bn_bwd/partial");
  }
  int gx = cdiv(HW, 256 * 8);
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(gx, B * C), dim3(256), 0, st, dy, x, y, stats, gamma, (const double*)ws, dx, dres, C, HW,
                     S, 1.f / ((float)B * HW), relu, frozen, have, dgamma, dbeta, beta, dbs);
  PRN_CHECK_LAUNCH("prn_bn_bwd/apply");
  return 0;
}

// The one-pass kernels fed with the K-split partial sums of the GEMM that produces their input (include/prn.h: prn_conv2d_fwd_partials): the
// separate sum launch of the producer and one read of its output disappear (models/backbone.py:56-66: conv1 -> bn1 in the forward, conv3's
// input gradient -> bn2's backward).  Only where prn_bn_kernel_kind(B, HW) == 1.
extern "C" int prn_bn_train_fwd_partials(const float* parts, int nparts, int64_t part_stride, float* x_out, float* stats, const float* gamma, const float* beta,
                                         const float* residual, float* y, float* running_mean, float* running_var, int B, int C, int HW, float eps,
                                         float momentum, int relu, void* stream) {
  PRN_REQUIRE(parts && x_out && stats && gamma && beta && y && B > 0 && C > 0 && HW > 0, "prn_bn_train_fwd_partials: bad arguments");
  PRN_REQUIRE(nparts >= 1 && nparts <= 64 && (nparts == 1 || part_stride >= (int64_t)B * C * HW) && (part_stride & 3) == 0,
              "prn_bn_train_fwd_partials: 1..64 partial sums, a multiple of four elements and at least B*C*HW apart");
  PRN_REQUIRE(bn_small_ok(B, HW), "prn_bn_train_fwd_partials: only for maps the one-pass kernel takes (prn_bn_kernel_kind(B, HW) == 1)");
  PRN_REQUIRE(((reinterpret_cast<uintptr_t>(parts) | reinterpret_cast<uintptr_t>(x_out) | reinterpret_cast<uintptr_t>(y)) & 15) == 0,
              "prn_bn_train_fwd_partials: tensors must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int nv = cdiv(B * HW / 4, 256);
  const int64_t ybs = (int64_t)C * HW;
#define PRN_BN_SMALL_FWD(NV_) hipLaunchKernelGGL((bn_small_fwd_kernel<NV_>), dim3(C), dim3(256), 0, st, parts, stats, gamma, beta, residual, y, \
                                                 running_mean, running_var, B, C, HW, eps, momentum, relu, ybs, nparts, part_stride, x_out)
  if (nv <= 3) PRN_BN_SMALL_FWD(3); else if (nv <= 6) PRN_BN_SMALL_FWD(6); else if (nv <= 10) PRN_BN_SMALL_FWD(10); else PRN_BN_SMALL_FWD(12);
#undef PRN_BN_SMALL_FWD
  PRN_CHECK_LAUNCH("prn_bn_train_fwd_partials");
  return 0;
}

extern "C" int prn_bn_bwd_partials(const float* dparts, int nparts, int64_t part_stride, const float* x, const float* y, const float* stats, const float* gamma,
                                   const float* beta, float* dx, float* dres, float* dgamma, float* dbeta, int B, int C, int HW, int relu, int frozen,
                                   void* stream) {
  PRN_REQUIRE(dparts && x && stats && gamma && dx && B > 0 && C > 0 && HW > 0, "prn_bn_bwd_partials: bad arguments");
  PRN_REQUIRE(nparts >= 1 && nparts <= 64 && (nparts == 1 || part_stride >= (int64_t)B * C * HW) && (part_stride & 3) == 0,
              "prn_bn_bwd_partials: 1..64 partial sums, a multiple of four elements and at least B*C*HW apart");
  PRN_REQUIRE(bn_small_ok(B, HW), "prn_bn_bwd_partials: only for maps the one-pass kernel takes (prn_bn_kernel_kind(B, HW) == 1)");
  PRN_REQUIRE((reinterpret_cast<uintptr_t>(dparts) & 15) == 0, "prn_bn_bwd_partials: partial sums must be 16-byte aligned");
  PRN_REQUIRE(!relu || y || (beta && !dres), "prn_bn_bwd_partials: relu needs the forward output, or beta (and no residual) to recompute its sign");
  if (relu) relu = y ? 1 : 2;
  hipStream_t st = (hipStream_t)stream;
  const int nv = cdiv(B * HW / 4, 256);
  const int64_t dbs = (int64_t)C * HW;
#define PRN_BN_SMALL_BWD(NV_) hipLaunchKernelGGL((bn_small_bwd_kernel<NV_>), dim3(C), dim3(256), 0, st, dparts, x, y, stats, gamma, beta, dx, dres, dgamma, \
                                                 dbeta, B, C, HW, relu, frozen, dbs, nparts, part_stride)
  if (nv <= 3) PRN_BN_SMALL_BWD(3); else if (nv <= 6) PRN_BN_SMALL_BWD(6); else if (nv <= 10) PRN_BN_SMALL_BWD(10); else PRN_BN_SMALL_BWD(12);
#undef PRN_BN_SMALL_BWD
  PRN_CHECK_LAUNCH("prn_bn_bwd_partials");
  return 0;
}

// ... and with the layer's OUTPUT (forward: y; backward: dx) leaving a second time as the Winograd input transform V = B^T . B [36][C][P4] that the 3x3 /
// pad-1 convolution behind (forward) / in front of (backward) the layer would compute from it (prn_winograd_input, PRN_IN_ZERO): that launch and its pass
// over the tensor disappear (models/backbone.py:57-60).  [B, C, H, W] with W % 4 == 0; otherwise the two calls above (nparts >= 1).
extern "C" int prn_bn_train_fwd_winograd(const float* parts, int nparts, int64_t part_stride, float* x_out, float* stats, const float* gamma, const float* beta,
                                         const float* residual, float* y, float* running_mean, float* running_var, float* V, int B, int C, int H, int W,
                                         float eps, float momentum, int relu, void* stream) {
  const int HW = H * W;
  PRN_REQUIRE(parts && stats && gamma && beta && y && V && B > 0 && C > 0 && H >= 5 && W >= 4 && (W & 3) == 0, "prn_bn_train_fwd_winograd: bad arguments");
  PRN_REQUIRE(nparts >= 1 && nparts <= 64 && (nparts == 1 || (x_out && part_stride >= (int64_t)B * C * HW)) && (part_stride & 3) == 0,
              "prn_bn_train_fwd_winograd: 1..64 partial sums (x_out required beyond one), a multiple of four elements and at least B*C*HW apart");
  PRN_REQUIRE(bn_small_ok(B, HW), "prn_bn_train_fwd_winograd: only for maps the one-pass kernel takes (prn_bn_kernel_kind(B, HW) == 1)");
  PRN_REQUIRE(((reinterpret_cast<uintptr_t>(parts) | reinterpret_cast<uintptr_t>(x_out) | reinterpret_cast<uintptr_t>(y)) & 15) == 0,
              "prn_bn_train_fwd_winograd: tensors must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int nv = cdiv(B * HW / 4, 256);
  const int64_t ybs = (int64_t)C * HW;
#define PRN_BN_SMALL_FWD(NV_) hipLaunchKernelGGL((bn_small_fwd_kernel<NV_, true>), dim3(C), dim3(256), 0, st, parts, stats, gamma, beta, residual, y, \
                                                 running_mean, running_var, B, C, HW, eps, momentum, relu, ybs, nparts, part_stride, x_out, V, H, W)
  if (nv <= 3) PRN_BN_SMALL_FWD(3); else if (nv <= 6) PRN_BN_SMALL_FWD(6); else if (nv <= 10) PRN_BN_SMALL_FWD(10); else PRN_BN_SMALL_FWD(12);
#undef PRN_BN_SMALL_FWD
  PRN_CHECK_LAUNCH("prn_bn_train_fwd_winograd");
  return 0;
}

extern "C" int prn_bn_bwd_winograd(const float* dparts, int nparts, int64_t part_stride, const float* x, const float* y, const float* stats, const float* gamma,
                                   const float* beta, float* dx, float* dres, float* dgamma, float* dbeta, float* V, int B, int C, int H, int W, int relu,
                                   int frozen, void* stream) {
  const int HW = H * W;
  PRN_REQUIRE(dparts && x && stats && gamma && dx && V && B > 0 && C > 0 && H >= 5 && W >= 4 && (W & 3) == 0, "prn_bn_bwd_winograd: bad arguments");
  PRN_REQUIRE(nparts >= 1 && nparts <= 64 && (nparts == 1 || part_stride >= (int64_t)B * C * HW) && (part_stride & 3) == 0,
              "prn_bn_bwd_winograd: 1..64 partial sums, a multiple of four elements and at least B*C*HW apart");
  PRN_REQUIRE(bn_small_ok(B, HW), "prn_bn_bwd_winograd: only for maps the one-pass kernel takes (prn_bn_kernel_kind(B, HW) == 1)");
  PRN_REQUIRE((reinterpret_cast<uintptr_t>(dparts) & 15) == 0, "prn_bn_bwd_winograd: the gradient must be 16-byte aligned");
  PRN_REQUIRE(!relu || y || (beta && !dres), "prn_bn_bwd_winograd: relu needs the forward output, or beta (and no residual) to recompute its sign");
  if (relu) relu = y ? 1 : 2;
  hipStream_t st = (hipStream_t)stream;
  const int nv = cdiv(B * HW / 4, 256);
  const int64_t dbs = (int64_t)C * HW;
#define PRN_BN_SMALL_BWD(NV_) hipLaunchKernelGGL((bn_small_bwd_kernel<NV_, true>), dim3(C), dim3(256), 0, st, dparts, x, y, stats, gamma, beta, dx, dres, dgamma, \
                                                 dbeta, B, C, HW, relu, frozen, dbs, nparts, part_stride, V, H, W)
  if (nv <= 3) PRN_BN_SMALL_BWD(3); else if (nv <= 6) PRN_BN_SMALL_BWD(6); else if (nv <= 10) PRN_BN_SMALL_BWD(10); else PRN_BN_SMALL_BWD(12);
#undef PRN_BN_SMALL_BWD
  PRN_CHECK_LAUNCH("prn_bn_bwd_winograd");
  return 0;
}

extern "C" int prn_gn_relu_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                               int B, int C, int HW, int G, float eps, void* stream) {
  PRN_REQUIRE(x && gamma && beta && y && stats && B > 0 && C > 0 && HW > 0 && G > 0 && C % G == 0, "prn_gn_relu_fwd: bad arguments");
  GnSeg sg; sg.nseg = 0;
  hipLaunchKernelGGL(gn_relu_fwd_kernel, dim3(B * G), dim3((C / G) * HW >= 32768 ? 1024 : 512), 0, (hipStream_t)stream, x, gamma, beta, y, stats, C, HW, G, eps, B * G, sg);
  PRN_CHECK_LAUNCH("prn_gn_relu_fwd");
  return 0;
}

static int gn_fill_seg(GnSeg& sg, int B, int C, int nseg, const int* hw) {
  PRN_REQUIRE(nseg >= 1 && nseg <= 6 && hw, "prn_gn_relu_*_ragged: 1..6 segments");
  int64_t off = 0;
  for (int s = 0; s < nseg; ++s) {
    PRN_REQUIRE(hw[s] > 0, "prn_gn_relu_*_ragged: empty segment");
    sg.hw[s] = hw[s]; sg.off[s] = (int)off;
    off += (int64_t)B * C * hw[s];
  }
  PRN_REQUIRE(off < (1LL << 31), "prn_gn_relu_*_ragged: batch too large");
  sg.nseg = nseg;
  return 0;
}

extern "C" int prn_gn_relu_fwd_ragged(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                                      int B, int C, int nseg, const int* hw, int G, float eps, void* stream) {
  PRN_REQUIRE(x && gamma && beta && y && stats && B > 0 && C > 0 && G > 0 && C % G == 0, "prn_gn_relu_fwd_ragged: bad arguments");
  GnSeg sg;
  if (int e = gn_fill_seg(sg, B, C, nseg, hw)) return e;
  hipLaunchKernelGGL(gn_relu_fwd_kernel, dim3(nseg * B * G), dim3(512), 0, (hipStream_t)stream, x, gamma, beta, y, stats, C, 0, G, eps, B * G, sg);
  PRN_CHECK_LAUNCH("prn_gn_relu_fwd_ragged");
  return 0;
}

extern "C" int prn_gn_relu_bwd(const float* dy, const float* x, const float* beta, const float* stats, const float* gamma,
                               float* dx, float* dgamma_part, float* dbeta_part, int B, int C, int HW, int G, void* stream) {
  PRN_REQUIRE(dy && x && beta && stats && gamma && dx && dgamma_part && dbeta_part && C % G == 0, "prn_gn_relu_bwd: bad arguments");
  GnSeg sg; sg.nseg = 0;
  hipLaunchKernelGGL(gn_relu_bwd_kernel, dim3(B * G), dim3((C / G) * HW >= 32768 ? 1024 : 512), 0, (hipStream_t)stream, dy, x, beta, stats, gamma, dx, dgamma_part, dbeta_part, C, HW, G,
                     B * G, sg);
  PRN_CHECK_LAUNCH("prn_gn_relu_bwd");
  return 0;
}

extern "C" int prn_gn_relu_bwd_ragged(const float* dy, const float* x, const float* beta, const float* stats, const float* gamma,
                                      float* dx, float* dgamma_part, float* dbeta_part, int B, int C, int nseg, const int* hw, int G,
                                      void* stream) {
  PRN_REQUIRE(dy && x && beta && stats && gamma && dx && dgamma_part && dbeta_part && C % G == 0, "prn_gn_relu_bwd_ragged: bad arguments");
  GnSeg sg;
  if (int e = gn_fill_seg(sg, B, C, nseg, hw)) return e;
  hipLaunchKernelGGL(gn_relu_bwd_kernel, dim3(nseg * B * G), dim3(512), 0, (hipStream_t)stream, dy, x, beta, stats, gamma, dx, dgamma_part, dbeta_part, C, 0,
                     G, B * G, sg);
  PRN_CHECK_LAUNCH("prn_gn_relu_bwd_ragged");
  return 0;
}

// out[z][n] = sum_r in[z][r][n] (r ascending: fixed order), z < nb: the per-image partials of the GroupNorm parameter gradients ([2][B][C] -> [2][C]),
// one launch instead of a framework reduction (so that the gradient can be issued on a side stream by pointer: planerecnet_amd.ops._small_param_grads).
namespace {
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int N, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t z = i / N;
  const int n = (int)(i - z * N);
  const float* p = in + z * R * N + n;
  float s = 0.f;
  for (int r = 0; r < R; ++r) s += p[(int64_t)r * N];
  out[i] = s;
}
}  // namespace

extern "C" int prn_sum_rows(const float* in, float* out, int nb, int R, int N, void* stream) {
  PRN_REQUIRE(in && out && nb > 0 && R > 0 && N > 0, "prn_sum_rows: bad arguments");
  const int64_t total = (int64_t)nb * N;
  hipLaunchKernelGGL(sum_rows_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, R, N, total);
  PRN_CHECK_LAUNCH("prn_sum_rows");
  return 0;
}

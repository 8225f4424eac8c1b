// Resampling family: bilinear resize (align_corners=False, PyTorch area_pixel source index rule) and
// MaxPool2d(3, 2, 1). HBM-bound gathers; one thread per output element, consecutive lanes along W.
#include "prn_common.h"

namespace {

struct Lerp { int i0, i1; float w0, w1; };

// PyTorch upsample_bilinear2d, align_corners=False: src = max(0, scale*(dst+0.5)-0.5), scale = in/out
__device__ __forceinline__ Lerp lerp_idx(int o, int in, float scale) {
  float s = scale * ((float)o + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  Lerp l;
  l.i0 = (int)s;
  if (l.i0 > in - 1) l.i0 = in - 1;
  l.i1 = l.i0 + (l.i0 < in - 1 ? 1 : 0);
  l.w1 = s - (float)l.i0;
  l.w0 = 1.f - l.w1;
  return l;
}

__global__ __launch_bounds__(256) void resize_fwd_kernel(const float* __restrict__ x, const float* __restrict__ addend, float* __restrict__ y, int64_t total,
                                                         int H, int W, int Ho, int Wo, float sh, float sw) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int wo = i % Wo, ho = (i / Wo) % Ho;
  const int64_t bc = i / ((int64_t)Wo * Ho);
  const Lerp a = lerp_idx(ho, H, sh), b = lerp_idx(wo, W, sw);
  const float* p = x + bc * (int64_t)H * W;
  const float v = a.w0 * (b.w0 * p[a.i0 * W + b.i0] + b.w1 * p[a.i0 * W + b.i1]) + a.w1 * (b.w0 * p[a.i1 * W + b.i0] + b.w1 * p[a.i1 * W + b.i1]);
  y[i] = addend ? v + addend[i] : v;
}

// The same for Wo % 4 == 0, four consecutive outputs per thread: the row interpolation once, the taps as scalar loads (neighbours in a row: L1 hits), the
// addend and the result as ONE aligned 16-byte access each -- the output side is most of this kernel's traffic.  Per-element arithmetic as above.
__global__ __launch_bounds__(256) void resize_fwd4_kernel(const float* __restrict__ x, const float* __restrict__ addend, float* __restrict__ y, int64_t total4,
                                                          int H, int W, int Ho, int Wo, float sh, float sw) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int Wo4 = Wo >> 2;
  const int wo = (int)(i % Wo4) * 4, ho = (i / Wo4) % Ho;
  const int64_t bc = i / ((int64_t)Wo4 * Ho);
  const Lerp a = lerp_idx(ho, H, sh);
  const float* p0 = x + bc * (int64_t)H * W + (int64_t)a.i0 * W;
  const float* p1 = x + bc * (int64_t)H * W + (int64_t)a.i1 * W;
  float t00[4], t01[4], t10[4], t11[4]; Lerp b[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    b[k] = lerp_idx(wo + k, W, sw);
    t00[k] = p0[b[k].i0]; t01[k] = p0[b[k].i1]; t10[k] = p1[b[k].i0]; t11[k] = p1[b[k].i1];
  }
  const int64_t o = (bc * Ho + ho) * (int64_t)Wo + wo;
  float v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = a.w0 * (b[k].w0 * t00[k] + b[k].w1 * t01[k]) + a.w1 * (b[k].w0 * t10[k] + b[k].w1 * t11[k]);
  if (addend) {
    const float4 ad = *reinterpret_cast<const float4*>(addend + o);
    v[0] = v[0] + ad.x; v[1] = v[1] + ad.y; v[2] = v[2] + ad.z; v[3] = v[3] + ad.w;
  }
  *reinterpret_cast<float4*>(y + o) = make_float4(v[0], v[1], v[2], v[3]);
}

// Adjoint of the bilinear resize in GATHER form (deterministic, no atomics): one thread per INPUT pixel sums the
// output pixels whose 2x2 footprint contains it.  Candidate outputs are bounded from the inverse of the source-index
// map and re-checked with the exact forward index computation, so border clamping is handled by construction.
__device__ __forceinline__ void cand_range(int i, int out, float scale, int& lo, int& hi) {
  const float inv = 1.f / scale;
  lo = (int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1;
  hi = (int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1;
  lo = lo < 0 ? 0 : lo;
  hi = hi > out - 1 ? out - 1 : hi;
  // the clamp of negative source coordinates maps every leading output onto input 0 / the trailing ones onto in-1
  if (i == 0) lo = 0;
}

__global__ __launch_bounds__(256) void resize_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ addend, float* __restrict__ dx, int64_t total,
                                                         int H, int W, int Ho, int Wo, float sh, float sw) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int w, h; int64_t bc;
  prn_idx3(i, W, H, w, h, bc);
  int hlo, hhi, wlo, whi;
  cand_range(h, Ho, sh, hlo, hhi);
  cand_range(w, Wo, sw, wlo, whi);
  if (h == H - 1) hhi = Ho - 1;
  if (w == W - 1) whi = Wo - 1;
  const float* p = dy + bc * (int64_t)Ho * Wo;
  float acc = 0.f;
  for (int oh = hlo; oh <= hhi; ++oh) {
    const Lerp a = lerp_idx(oh, H, sh);
    const float wh = (a.i0 == h ? a.w0 : 0.f) + (a.i1 == h ? a.w1 : 0.f);
    if (wh == 0.f) continue;
    float row = 0.f;
    for (int ow = wlo; ow <= whi; ++ow) {
      const Lerp b = lerp_idx(ow, W, sw);
      const float ww = (b.i0 == w ? b.w0 : 0.f) + (b.i1 == w ? b.w1 : 0.f);
      row += ww * p[(int64_t)oh * Wo + ow];
    }
    acc += wh * row;
  }
  dx[i] = addend ? acc + addend[i] : acc;
}

// ---- exact factor-2 cases (FPN bottom-up path and split_feats: x0.5; mask-head levels and the depth loss: x2).  Same
// arithmetic, in the same order, as the generic kernels -- every weight is 0.25 / 0.5 / 0.75 / 1 exactly -- without the
// per-element index search: the generic adjoint ran at 0.7 TB/s on the 8x256x120x160 FPN gradient (285 us).
__global__ __launch_bounds__(256) void resize_down2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t total, int Ho, int Wo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int wo = i % Wo, ho = (i / Wo) % Ho;
  const int64_t bc = i / ((int64_t)Wo * Ho);
  const float* p = x + (bc * 2 * Ho + 2 * ho) * (int64_t)(2 * Wo) + 2 * wo;
  const float2 r0 = *reinterpret_cast<const float2*>(p), r1 = *reinterpret_cast<const float2*>(p + 2 * Wo);
  y[i] = 0.5f * (0.5f * r0.x + 0.5f * r0.y) + 0.5f * (0.5f * r1.x + 0.5f * r1.y);
}

__global__ __launch_bounds__(256) void resize_down2_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ addend, float* __restrict__ dx, int64_t total, int Ho, int Wo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one thread per PAIR of input pixels (2h', 2wo), (2h', 2wo+1)
  if (i >= total) return;
  const int wo = i % Wo, h = (i / Wo) % (2 * Ho);
  const int64_t bc = i / ((int64_t)Wo * 2 * Ho);
  const float g = 0.5f * (0.5f * dy[(bc * Ho + (h >> 1)) * (int64_t)Wo + wo]);
  const int64_t o = (bc * 2 * Ho + h) * (int64_t)(2 * Wo) + 2 * wo;
  float2 r = make_float2(g, g);
  if (addend) { const float2 a = *reinterpret_cast<const float2*>(addend + o); r.x += a.x; r.y += a.y; }      // another gradient of the same tensor
  *reinterpret_cast<float2*>(dx + o) = r;
}

// adjoint of the x2 upsample: input (i, j) is touched by output rows 2i-1 .. 2i+2 with weights .25 .75 .75 .25 (1.0 where the
// border clamp folds two of them together), same for columns
__device__ __forceinline__ float up2_weight(int o, int i, int in) {         // weight of output index o on input index i
  const int base = o >> 1;                                                   // o = 2*base + a
  if (o & 1) return (base == i ? 0.75f : 0.f) + ((base + 1 > in - 1 ? in - 1 : base + 1) == i ? 0.25f : 0.f);
  if (o == 0) return i == 0 ? 1.f : 0.f;                                     // source coordinate clamped to 0
  return (base - 1 == i ? 0.25f : 0.f) + (base == i ? 0.75f : 0.f);
}

__global__ __launch_bounds__(256) void resize_up2_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ addend, float* __restrict__ dx, int64_t total, int H, int W) {
  PRN_EPT_BEGIN(total)
    const int w = i % W, h = (i / W) % H;
    const int64_t bc = i / ((int64_t)W * H);
    const int Ho = 2 * H, Wo = 2 * W;
    const float* p = dy + bc * (int64_t)Ho * Wo;
    // Straight-line 4 x 4 footprint (rows 2h-1 .. 2h+2, columns 2w-1 .. 2w+2): sixteen loads in flight per element.  A position outside the
    // map gets weight 0 and a clamped address -- it adds +-0 to a partial sum that is never -0, i.e. nothing: the same value, summed in the same
    // order, as the loop over the clipped ranges it replaces.
    float wr[4], wc[4]; int orow[4], ocol[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int oh = 2 * h - 1 + a, ow = 2 * w - 1 + a;
      wr[a] = (oh >= 0 && oh <= Ho - 1) ? up2_weight(oh, h, H) : 0.f;
      wc[a] = (ow >= 0 && ow <= Wo - 1) ? up2_weight(ow, w, W) : 0.f;
      orow[a] = min(max(oh, 0), Ho - 1); ocol[a] = min(max(ow, 0), Wo - 1);
    }
    float t[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) t[a][b] = p[(int64_t)orow[a] * Wo + ocol[b]];
    // taps with weight 0 (outside the map, or an in-range position the interpolation does not use) contribute NOTHING, as in the clipped loop
    // and in ATen's backward: selected, not multiplied, so that an inf / NaN gradient next to the border does not leak into dx through 0 * inf
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) t[a][b] = (wr[a] != 0.f && wc[b] != 0.f) ? t[a][b] : 0.f;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float row = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) row += wc[b] * t[a][b];
      acc += wr[a] * row;
    }
    v_[k_] = addend ? acc + addend[i] : acc;
  PRN_EPT_END(total, dx)
}

// arg (optional): position r*3+s of the maximum inside the window (first maximum in scan order, ATen's
// max_pool2d_with_indices rule; NaN wins like in ATen), kept as one byte per output for the backward.
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ arg,
                                                          int64_t total, int H, int W, int Ho, int Wo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int wo = i % Wo, ho = (i / Wo) % Ho;
  const int64_t bc = i / ((int64_t)Wo * Ho);
  const float* p = x + bc * (int64_t)H * W;
  float m = -INFINITY;
  int am = 255;
  for (int r = 0; r < 3; ++r) {
    const int h = ho * 2 - 1 + r;
    if ((unsigned)h >= (unsigned)H) continue;
    for (int s = 0; s < 3; ++s) {
      const int w = wo * 2 - 1 + s;
      if ((unsigned)w >= (unsigned)W) continue;
      const float v = p[h * W + w];
      if (v > m || v != v || am == 255) { m = v; am = r * 3 + s; }
    }
  }
  y[i] = m;
  if (arg) arg[i] = (unsigned char)am;
}

// gradient goes to the recorded maximum; GATHER form: input (h, w) lies in at most four windows -- window row ho = (h+1-r)/2
// for the r in {0,1,2} with the parity of h+1 -- and takes dy of those whose recorded position is (r, s).  No atomics, no
// zero fill, deterministic (the scatter version: 266 us + a 157 MB fill at the stem).
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const unsigned char* __restrict__ arg, const float* __restrict__ dy,
                                                          float* __restrict__ dx, int64_t total, int H, int W, int Ho, int Wo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int w = i % W, h = (i / W) % H;
  const int64_t bc = i / ((int64_t)W * H);
  const int64_t ob = bc * (int64_t)Ho * Wo;
  float acc = 0.f;
  for (int r = (h + 1) & 1; r < 3; r += 2) {
    const int ho = (h + 1 - r) >> 1;
    if (ho < 0 || ho >= Ho) continue;
    for (int s = (w + 1) & 1; s < 3; s += 2) {
      const int wo = (w + 1 - s) >> 1;
      if (wo < 0 || wo >= Wo) continue;
      const int64_t o = ob + (int64_t)ho * Wo + wo;
      if (arg[o] == r * 3 + s) acc += dy[o];
    }
  }
  dx[i] = acc;
}

// The same for four consecutive input pixels per thread (W % 4 == 0, 16-byte aligned dx): the four pixels (h, w0 .. w0 + 3), w0 even, lie in the
// window columns wo = w0/2 .. w0/2 + 2 of one or two window rows -- at most six (arg, dy) pairs for one dwordx4 store.  The scalar form
// moved 157 MB of dx with one dword store per thread (217 us at the stem, 0.7 TB/s).
__global__ __launch_bounds__(256) void maxpool_bwd4_kernel(const unsigned char* __restrict__ arg, const float* __restrict__ dy,
                                                           float* __restrict__ dx, int64_t total4, int H, int W, int Ho, int Wo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int W4 = W >> 2;
  const int w0 = (int)(i % W4) * 4, h = (int)((i / W4) % H);
  const int64_t bc = i / ((int64_t)W4 * H);
  const int64_t ob = bc * (int64_t)Ho * Wo;
  const int wb = w0 >> 1;                                   // window columns wb, wb + 1, wb + 2
  float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
  for (int r = (h + 1) & 1; r < 3; r += 2) {
    const int ho = (h + 1 - r) >> 1;
    if (ho < 0 || ho >= Ho) continue;
    const int64_t rowo = ob + (int64_t)ho * Wo;
    int a[3]; float g[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const bool ok = wb + q < Wo;
      a[q] = ok ? (int)arg[rowo + wb + q] : 255;
      g[q] = ok ? dy[rowo + wb + q] : 0.f;
    }
    const int base = r * 3;
    // input column w: even w = 2 wo -> position s = 1 of window wo;  odd w = 2 wo + 1 -> s = 2 of window wo and s = 0 of window wo + 1
    // (same order of additions as the scalar kernel -- s = 0 before s = 2 --: bit-identical results)
    if (a[0] == base + 1) o0 += g[0];                       // w0     : s = 1 of window wb
    if (a[1] == base + 0) o1 += g[1];                       // w0 + 1 : s = 0 of window wb + 1
    if (a[0] == base + 2) o1 += g[0];                       //          s = 2 of window wb
    if (a[1] == base + 1) o2 += g[1];                       // w0 + 2 : s = 1 of window wb + 1
    if (a[2] == base + 0) o3 += g[2];                       // w0 + 3 : s = 0 of window wb + 2
    if (a[1] == base + 2) o3 += g[1];                       //          s = 2 of window wb + 1
  }
  *reinterpret_cast<float4*>(dx + i * 4) = make_float4(o0, o1, o2, o3);
}

}  // namespace

extern "C" int prn_resize_bilinear_fwd(const float* x, float* y, int BC, int H, int W, int Ho, int Wo, void* stream) {
  return prn_resize_bilinear_add_fwd(x, nullptr, y, BC, H, W, Ho, Wo, stream);
}

extern "C" int prn_resize_bilinear_add_fwd(const float* x, const float* addend, float* y, int BC, int H, int W, int Ho, int Wo, void* stream) {
  PRN_REQUIRE(x && y && BC > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "prn_resize_bilinear_fwd: bad arguments");
  const int64_t n = (int64_t)BC * Ho * Wo;
  if (addend == nullptr && H == 2 * Ho && W == 2 * Wo && (reinterpret_cast<uintptr_t>(x) & 7) == 0) {
    for (int r = PRN_REPS(16); r > 0; --r) hipLaunchKernelGGL(resize_down2_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, Ho, Wo);
    PRN_CHECK_LAUNCH("prn_resize_bilinear_fwd/down2");
    return 0;
  }
  if ((Wo & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(addend) & 15) == 0)
    hipLaunchKernelGGL(resize_fwd4_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, addend, y, n / 4, H, W, Ho, Wo, (float)H / Ho, (float)W / Wo);
  else
    hipLaunchKernelGGL(resize_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, addend, y, n, H, W, Ho, Wo, (float)H / Ho, (float)W / Wo);
  PRN_CHECK_LAUNCH("prn_resize_bilinear_fwd");
  return 0;
}

extern "C" int prn_resize_bilinear_bwd(const float* dy, float* dx, int BC, int H, int W, int Ho, int Wo, void* stream) {
  return prn_resize_bilinear_bwd_add(dy, nullptr, dx, BC, H, W, Ho, Wo, stream);
}

extern "C" int prn_resize_bilinear_bwd_add(const float* dy, const float* addend, float* dx, int BC, int H, int W, int Ho, int Wo, void* stream) {
  PRN_REQUIRE(dy && dx && BC > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "prn_resize_bilinear_bwd: bad arguments");
  const int64_t n = (int64_t)BC * H * W;
  if (H == 2 * Ho && W == 2 * Wo && ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(addend)) & 7) == 0) {
    for (int r = PRN_REPS(16); r > 0; --r) hipLaunchKernelGGL(resize_down2_bwd_kernel, dim3(cdiv(n / 2, 256)), dim3(256), 0, (hipStream_t)stream, dy, addend, dx, n / 2, Ho, Wo);
    PRN_CHECK_LAUNCH("prn_resize_bilinear_bwd/down2");
    return 0;
  }
  if (Ho == 2 * H && Wo == 2 * W) {
    hipLaunchKernelGGL(resize_up2_bwd_kernel, dim3(prn_ept_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy, addend, dx, n, H, W);
    PRN_CHECK_LAUNCH("prn_resize_bilinear_bwd/up2");
    return 0;
  }
  hipLaunchKernelGGL(resize_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, addend, dx, n, H, W, Ho, Wo, (float)H / Ho, (float)W / Wo);
  PRN_CHECK_LAUNCH("prn_resize_bilinear_bwd");
  return 0;
}

extern "C" int prn_maxpool3s2_fwd(const float* x, float* y, unsigned char* arg, int BC, int H, int W, int Ho, int Wo, void* stream) {
  PRN_REQUIRE(x && y && BC > 0 && Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1, "prn_maxpool3s2_fwd: bad arguments");
  const int64_t n = (int64_t)BC * Ho * Wo;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, arg, n, H, W, Ho, Wo);
  PRN_CHECK_LAUNCH("prn_maxpool3s2_fwd");
  return 0;
}

extern "C" int prn_maxpool3s2_bwd(const unsigned char* arg, const float* dy, float* dx, int BC, int H, int W, int Ho, int Wo, void* stream) {
  PRN_REQUIRE(arg && dy && dx && BC > 0, "prn_maxpool3s2_bwd: bad arguments");
  const int64_t n = (int64_t)BC * H * W;
  if ((W & 3) == 0 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0)
    hipLaunchKernelGGL(maxpool_bwd4_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, arg, dy, dx, n / 4, H, W, Ho, Wo);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, arg, dy, dx, n, H, W, Ho, Wo);
  PRN_CHECK_LAUNCH("prn_maxpool3s2_bwd");
  return 0;
}

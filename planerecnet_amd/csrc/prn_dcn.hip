// DCNv2 sampling: offset gather + bilinear sample + modulation, forward and backward.
// One thread owns one (b, tap, ho, wo): the sampling position, the four corner weights and the corner
// validity are computed once and reused for every input channel; the channel loop streams x planes with the
// wave's 64 consecutive wo positions touching neighbouring addresses, and writes the column tensor fully
// coalesced.  HBM-bound (no contraction here; the [Cout x 9Cin] GEMM runs on the MFMA conv kernel).
#include "prn_common.h"

namespace {

struct Tap {
  int i00, i01, i10, i11;   // corner offsets inside one channel plane (valid or 0)
  float w00, w01, w10, w11; // corner weights (0 when the corner is outside)
  float gy00, gy01, gy10, gy11, gx00, gx01, gx10, gx11;  // d(weight)/dy, d(weight)/dx
  float mod;                // 2*sigmoid(raw)
  float sig;                // sigmoid(raw)
  bool pass_y, pass_x;      // clamp passes gradient
};

// Sampling geometry of torchvision's deform_conv2d (see oracle/dcn_ref.py for the restated rule).
__device__ __forceinline__ Tap make_tap(const float* __restrict__ om, int b, int k, int ho, int wo, int Ho, int Wo,
                                        int H, int W, int stride, float maxoff, bool need_grad) {
  Tap t;
  const size_t plane = (size_t)Ho * Wo, pix = (size_t)ho * Wo + wo;
  const float* ob = om + (size_t)b * 27 * plane;
  const float ry = ob[(2 * k) * plane + pix], rx = ob[(2 * k + 1) * plane + pix], rm = ob[(18 + k) * plane + pix];
  const float dy = fminf(fmaxf(ry, -maxoff), maxoff), dx = fminf(fmaxf(rx, -maxoff), maxoff);
  t.pass_y = (ry >= -maxoff) && (ry <= maxoff);
  t.pass_x = (rx >= -maxoff) && (rx <= maxoff);
  t.sig = 1.f / (1.f + expf(-rm));
  t.mod = 2.f * t.sig;
  const int ki = k / 3, kj = k - ki * 3;
  const float y = (float)(ho * stride - 1 + ki) + dy, x = (float)(wo * stride - 1 + kj) + dx;
  const bool inside = (y > -1.f) && (y < (float)H) && (x > -1.f) && (x < (float)W);
  const float fy = floorf(y), fx = floorf(x);
  const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
  const float ly = y - fy, lx = x - fx, hy = 1.f - ly, hx = 1.f - lx;
  const bool vy0 = inside && y0 >= 0, vy1 = inside && y1 <= H - 1, vx0 = x0 >= 0, vx1 = x1 <= W - 1;
  const bool v00 = vy0 && vx0, v01 = vy0 && vx1, v10 = vy1 && vx0, v11 = vy1 && vx1;
  t.i00 = v00 ? y0 * W + x0 : 0; t.i01 = v01 ? y0 * W + x1 : 0;
  t.i10 = v10 ? y1 * W + x0 : 0; t.i11 = v11 ? y1 * W + x1 : 0;
  t.w00 = v00 ? hy * hx : 0.f; t.w01 = v01 ? hy * lx : 0.f;
  t.w10 = v10 ? ly * hx : 0.f; t.w11 = v11 ? ly * lx : 0.f;
  if (need_grad) {
    t.gy00 = v00 ? -hx : 0.f; t.gy01 = v01 ? -lx : 0.f; t.gy10 = v10 ? hx : 0.f; t.gy11 = v11 ? lx : 0.f;
    t.gx00 = v00 ? -hy : 0.f; t.gx01 = v01 ? hy : 0.f; t.gx10 = v10 ? -ly : 0.f; t.gx11 = v11 ? ly : 0.f;
  }
  return t;
}

// forward: blockIdx.y splits the channel loop so that the launch has several blocks per CU even on 30x40 maps
__global__ __launch_bounds__(256) void dcn_sample_kernel(const float* __restrict__ x, const float* __restrict__ om,
                                                         float* __restrict__ cols, int B, int C, int H, int W, int Ho,
                                                         int Wo, int stride, float maxoff) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t plane = (int64_t)Ho * Wo;
  if (gid >= (int64_t)B * 9 * plane) return;
  const int pix = gid % plane, k = (gid / plane) % 9, b = gid / (9 * plane);
  const int ho = pix / Wo, wo = pix - ho * Wo;
  const Tap t = make_tap(om, b, k, ho, wo, Ho, Wo, H, W, stride, maxoff, false);
  const int cper = (C + gridDim.y - 1) / gridDim.y;
  const int c0 = blockIdx.y * cper, c1 = min(C, c0 + cper);
  const size_t HW = (size_t)H * W;
  const float* xp = x + ((size_t)b * C + c0) * HW;
  float* cp = cols + (((size_t)b * C + c0) * 9 + k) * plane + pix;
#pragma unroll 4
  for (int c = c0; c < c1; ++c) {
    const float v = t.w00 * xp[t.i00] + t.w01 * xp[t.i01] + t.w10 * xp[t.i10] + t.w11 * xp[t.i11];
    *cp = t.mod * v;
    xp += HW;
    cp += 9 * plane;
  }
}

// backward, part 1: gradients of the raw offset / modulator maps. Thread per (b, tap, pixel), channel loop split over
// blockIdx.y into fixed-order partials (no atomics): part[g][b][27][Ho*Wo].
__global__ __launch_bounds__(256) void dcn_dom_partial_kernel(const float* __restrict__ x, const float* __restrict__ om,
                                                              const float* __restrict__ dcols, float* __restrict__ part, int B,
                                                              int C, int H, int W, int Ho, int Wo, int stride, float maxoff) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t plane = (int64_t)Ho * Wo;
  if (gid >= (int64_t)B * 9 * plane) return;
  const int pix = gid % plane, k = (gid / plane) % 9, b = gid / (9 * plane);
  const int ho = pix / Wo, wo = pix - ho * Wo;
  const Tap t = make_tap(om, b, k, ho, wo, Ho, Wo, H, W, stride, maxoff, true);
  const int cper = (C + gridDim.y - 1) / gridDim.y;
  const int c0 = blockIdx.y * cper, c1 = min(C, c0 + cper);
  const size_t HW = (size_t)H * W;
  const float* xp = x + ((size_t)b * C + c0) * HW;
  const float* dp = dcols + (((size_t)b * C + c0) * 9 + k) * plane + pix;
  float gy = 0.f, gx = 0.f, gm = 0.f;
#pragma unroll 4
  for (int c = c0; c < c1; ++c) {
    const float g = *dp;
    const float x00 = xp[t.i00], x01 = xp[t.i01], x10 = xp[t.i10], x11 = xp[t.i11];
    gm += g * (t.w00 * x00 + t.w01 * x01 + t.w10 * x10 + t.w11 * x11);
    gy += g * (t.gy00 * x00 + t.gy01 * x01 + t.gy10 * x10 + t.gy11 * x11);
    gx += g * (t.gx00 * x00 + t.gx01 * x01 + t.gx10 * x10 + t.gx11 * x11);
    xp += HW;
    dp += 9 * plane;
  }
  float* ob = part + ((size_t)blockIdx.y * B + b) * 27 * plane + pix;
  ob[(2 * k) * plane] = t.pass_y ? gy * t.mod : 0.f;
  ob[(2 * k + 1) * plane] = t.pass_x ? gx * t.mod : 0.f;
  ob[(18 + k) * plane] = gm * 2.f * t.sig * (1.f - t.sig);
}

__global__ void dcn_dom_final_kernel(const float* __restrict__ part, float* __restrict__ d_om, int64_t n, int G) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int g = 0; g < G; ++g) s += part[(size_t)g * n + i];
  d_om[i] = s;
}

// backward, part 2: d-input.  The scatter of every sampling point onto its four corners is privatised in LDS: a block
// owns CG whole channel planes of one image (CG*H*W floats of LDS), walks all 9*Ho*Wo sampling points, accumulates with
// LDS atomics and finally streams the planes out with plain coalesced stores -- no global atomics, no zero-fill of dx.
// 1024 threads per block and all CG column loads issued before the first atomic keep enough global loads in flight.
template <int CG>
__global__ __launch_bounds__(1024) void dcn_dx_lds_kernel(const float* __restrict__ om, const float* __restrict__ dcols,
                                                          float* __restrict__ dx, int B, int C, int H, int W, int Ho, int Wo,
                                                          int stride, float maxoff) {
  extern __shared__ float planes[];                  // [CG][H*W]
  const int HW = H * W;
  const int64_t plane = (int64_t)Ho * Wo;
  const int groups = (C + CG - 1) / CG;
  const int b = blockIdx.x / groups, c0 = (blockIdx.x % groups) * CG;
  const int cg = min(CG, C - c0);
  for (int i = threadIdx.x; i < CG * HW; i += 1024) planes[i] = 0.f;
  __syncthreads();
  const int ntap = 9 * (int)plane;
  for (int tix = threadIdx.x; tix < ntap; tix += 1024) {
    const int k = tix / (int)plane, pix = tix - k * (int)plane;
    const int ho = pix / Wo, wo = pix - ho * Wo;
    const float* dp = dcols + (((size_t)b * C + c0) * 9 + k) * plane + pix;
    float g[CG];
#pragma unroll
    for (int cc = 0; cc < CG; ++cc) g[cc] = dp[(size_t)(cc < cg ? cc : 0) * 9 * plane];
    const Tap t = make_tap(om, b, k, ho, wo, Ho, Wo, H, W, stride, maxoff, false);
#pragma unroll
    for (int cc = 0; cc < CG; ++cc) {
      const float gm = (cc < cg) ? g[cc] * t.mod : 0.f;
      float* pl = planes + cc * HW;
      atomicAdd(pl + t.i00, gm * t.w00);             // invalid corners carry weight 0 and index 0: harmless adds
      atomicAdd(pl + t.i01, gm * t.w01);
      atomicAdd(pl + t.i10, gm * t.w10);
      atomicAdd(pl + t.i11, gm * t.w11);
    }
  }
  __syncthreads();
  float* out = dx + ((size_t)b * C + c0) * HW;
  for (int i = threadIdx.x; i < cg * HW; i += 1024) out[i] = planes[i];
}

template <int CG>
void launch_dx(const float* om, const float* dcols, float* dx, int B, int C, int H, int W, int Ho, int Wo, int stride, float maxoff,
               hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dcn_dx_lds_kernel<CG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((dcn_dx_lds_kernel<CG>), dim3(B * cdiv(C, CG)), dim3(1024), (size_t)CG * H * W * 4, st, om, dcols, dx, B, C, H, W, Ho, Wo,
                     stride, maxoff);
}

}  // namespace

static int channel_groups(int C, int64_t threads) {
  int g = (int)(1 + (256 * 8 * 256) / (threads > 0 ? threads : 1));     // aim for ~8 blocks of 256 threads per CU
  if (g > 8) g = 8;
  while (g > 1 && C % g) --g;
  return g < 1 ? 1 : g;
}

extern "C" int prn_dcn_sample(const float* x, const float* om, float* cols, int B, int C, int H, int W, int Ho, int Wo,
                              int stride, float max_offset, void* stream) {
  PRN_REQUIRE(x && om && cols && B > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "prn_dcn_sample: bad arguments");
  PRN_REQUIRE(Ho == (H + 2 - 3) / stride + 1 && Wo == (W + 2 - 3) / stride + 1, "prn_dcn_sample: output size mismatch");
  const int64_t n = (int64_t)B * 9 * Ho * Wo;
  hipLaunchKernelGGL(dcn_sample_kernel, dim3(cdiv(n, 256), channel_groups(C, n)), dim3(256), 0, (hipStream_t)stream, x, om, cols, B, C, H, W, Ho,
                     Wo, stride, max_offset);
  PRN_CHECK_LAUNCH("prn_dcn_sample");
  return 0;
}

extern "C" int64_t prn_dcn_sample_bwd_ws_bytes(int B, int C, int Ho, int Wo) {
  return (int64_t)channel_groups(C, (int64_t)B * 9 * Ho * Wo) * B * 27 * Ho * Wo * 4;
}

extern "C" int prn_dcn_sample_bwd(const float* x, const float* om, const float* dcols, float* dx, float* d_om, void* ws,
                                  int B, int C, int H, int W, int Ho, int Wo, int stride, float max_offset, void* stream) {
  PRN_REQUIRE(x && om && dcols && dx && d_om && ws && B > 0 && C > 0, "prn_dcn_sample_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = (int64_t)B * 9 * Ho * Wo;
  const int G = channel_groups(C, n);
  hipLaunchKernelGGL(dcn_dom_partial_kernel, dim3(cdiv(n, 256), G), dim3(256), 0, st, x, om, dcols, (float*)ws, B, C, H, W, Ho, Wo, stride, max_offset);
  PRN_CHECK_LAUNCH("prn_dcn_sample_bwd/d_om partial");
  const int64_t nom = (int64_t)B * 27 * Ho * Wo;
  hipLaunchKernelGGL(dcn_dom_final_kernel, dim3(cdiv(nom, 256)), dim3(256), 0, st, (const float*)ws, d_om, nom, G);
  PRN_CHECK_LAUNCH("prn_dcn_sample_bwd/d_om final");
  // channel planes per block: the largest power of two <= 16 that fits 64 KB of LDS
  const int HW = H * W;
  PRN_REQUIRE((int64_t)HW * 4 <= 150 * 1024, "prn_dcn_sample_bwd: a %dx%d plane does not fit LDS", H, W);
  const int fit = (64 * 1024) / (HW * 4);
  if (fit >= 16) launch_dx<16>(om, dcols, dx, B, C, H, W, Ho, Wo, stride, max_offset, st);
  else if (fit >= 8) launch_dx<8>(om, dcols, dx, B, C, H, W, Ho, Wo, stride, max_offset, st);
  else if (fit >= 4) launch_dx<4>(om, dcols, dx, B, C, H, W, Ho, Wo, stride, max_offset, st);
  else if (fit >= 2) launch_dx<2>(om, dcols, dx, B, C, H, W, Ho, Wo, stride, max_offset, st);
  else launch_dx<1>(om, dcols, dx, B, C, H, W, Ho, Wo, stride, max_offset, st);
  PRN_CHECK_LAUNCH("prn_dcn_sample_bwd/dx");
  return 0;
}

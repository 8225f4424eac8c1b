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

__global__ __launch_bounds__(256) void dcn_sample_kernel(const float* __restrict__ x, const float* __restrict__ om,
                                                         float* __restrict__ cols, int B, int C, int H, int W, int Ho,
                                                         int Wo, int stride, float maxoff) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t plane = (int64_t)Ho * Wo;
  if (gid >= (int64_t)B * 9 * plane) return;
  const int pix = gid % plane, k = (gid / plane) % 9, b = gid / (9 * plane);
  const int ho = pix / Wo, wo = pix - ho * Wo;
  const Tap t = make_tap(om, b, k, ho, wo, Ho, Wo, H, W, stride, maxoff, false);
  const float* xp = x + (size_t)b * C * H * W;
  float* cp = cols + ((size_t)b * C * 9 + k) * plane + pix;
  const size_t HW = (size_t)H * W;
#pragma unroll 4
  for (int c = 0; c < C; ++c) {
    const float v = t.w00 * xp[t.i00] + t.w01 * xp[t.i01] + t.w10 * xp[t.i10] + t.w11 * xp[t.i11];
    *cp = t.mod * v;
    xp += HW;
    cp += 9 * plane;
  }
}

__global__ __launch_bounds__(256) void dcn_sample_bwd_kernel(const float* __restrict__ x, const float* __restrict__ om,
                                                             const float* __restrict__ dcols, float* __restrict__ dx,
                                                             float* __restrict__ d_om, int B, int C, int H, int W, int Ho,
                                                             int Wo, int stride, float maxoff) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t plane = (int64_t)Ho * Wo;
  if (gid >= (int64_t)B * 9 * plane) return;
  const int pix = gid % plane, k = (gid / plane) % 9, b = gid / (9 * plane);
  const int ho = pix / Wo, wo = pix - ho * Wo;
  const Tap t = make_tap(om, b, k, ho, wo, Ho, Wo, H, W, stride, maxoff, true);
  const size_t HW = (size_t)H * W;
  const float* xp = x + (size_t)b * C * HW;
  float* dxp = dx + (size_t)b * C * HW;
  const float* dp = dcols + ((size_t)b * C * 9 + k) * plane + pix;
  float gy = 0.f, gx = 0.f, gm = 0.f;
  for (int c = 0; c < C; ++c) {
    const float g = *dp;
    const float x00 = xp[t.i00], x01 = xp[t.i01], x10 = xp[t.i10], x11 = xp[t.i11];
    gm += g * (t.w00 * x00 + t.w01 * x01 + t.w10 * x10 + t.w11 * x11);
    const float gmod = g * t.mod;
    gy += gmod * (t.gy00 * x00 + t.gy01 * x01 + t.gy10 * x10 + t.gy11 * x11);
    gx += gmod * (t.gx00 * x00 + t.gx01 * x01 + t.gx10 * x10 + t.gx11 * x11);
    if (t.w00 != 0.f) atomicAdd(dxp + t.i00, gmod * t.w00);
    if (t.w01 != 0.f) atomicAdd(dxp + t.i01, gmod * t.w01);
    if (t.w10 != 0.f) atomicAdd(dxp + t.i10, gmod * t.w10);
    if (t.w11 != 0.f) atomicAdd(dxp + t.i11, gmod * t.w11);
    xp += HW;
    dxp += HW;
    dp += 9 * plane;
  }
  float* ob = d_om + (size_t)b * 27 * plane + pix;
  ob[(2 * k) * plane] = t.pass_y ? gy : 0.f;
  ob[(2 * k + 1) * plane] = t.pass_x ? gx : 0.f;
  ob[(18 + k) * plane] = gm * 2.f * t.sig * (1.f - t.sig);
}

}  // namespace

extern "C" int prn_dcn_sample(const float* x, const float* om, float* cols, int B, int C, int H, int W, int Ho, int Wo,
                              int stride, float max_offset, void* stream) {
  PRN_REQUIRE(x && om && cols && B > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "prn_dcn_sample: bad arguments");
  PRN_REQUIRE(Ho == (H + 2 - 3) / stride + 1 && Wo == (W + 2 - 3) / stride + 1, "prn_dcn_sample: output size mismatch");
  const int64_t n = (int64_t)B * 9 * Ho * Wo;
  hipLaunchKernelGGL(dcn_sample_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, om, cols, B, C, H, W, Ho, Wo, stride, max_offset);
  PRN_CHECK_LAUNCH("prn_dcn_sample");
  return 0;
}

extern "C" int prn_dcn_sample_bwd(const float* x, const float* om, const float* dcols, float* dx, float* d_om,
                                  int B, int C, int H, int W, int Ho, int Wo, int stride, float max_offset, void* stream) {
  PRN_REQUIRE(x && om && dcols && dx && d_om && B > 0 && C > 0, "prn_dcn_sample_bwd: bad arguments");
  const int64_t n = (int64_t)B * 9 * Ho * Wo;
  hipLaunchKernelGGL(dcn_sample_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, om, dcols, dx, d_om, B, C, H, W, Ho, Wo, stride, max_offset);
  PRN_CHECK_LAUNCH("prn_dcn_sample_bwd");
  return 0;
}

// Helpers of the sub-pixel form of "Upsample(x2, nearest) -> ReflectionPad2d(1) -> Conv3x3" (PRN_IN_UP2_PHASE).
//
// On a nearest-x2 upsampled map the reflect border equals a replicate border of the source, and the three taps of a row
// (column) of the 3x3 kernel fall on only two source rows (columns): output row 2i sees source rows (i-1, i) with weights
// (w0, w1+w2), output row 2i+1 sees (i, i+1) with (w0+w1, w2).  So the layer is four 2x2 convolutions of the SOURCE map,
// one per output phase -- 16 instead of 36 multiply-adds per source pixel, channel pair and output phase group.  Its input
// gradient is a 4x4 stride-2 convolution of dy (rows 2r-1 .. 2r+2 reach padded source row r), followed by folding the
// replicate border; its weight gradient is taken per phase against the de-interleaved dy and mapped back to 3x3.
// All kernels here are tiny permutations / sums; the contractions run on the MFMA kernels of prn_conv.hip.
#include "prn_common.h"

namespace {

// taps of the 3x3 kernel that land on window position u of phase p:  (p,u) = (0,0):{0}  (0,1):{1,2}  (1,0):{0,1}  (1,1):{2}
__device__ __forceinline__ void phase_taps(int p, int u, int& lo, int& hi) {
  lo = (p == 0) ? (u == 0 ? 0 : 1) : (u == 0 ? 0 : 2);
  hi = (p == 0) ? (u == 0 ? 0 : 2) : (u == 0 ? 1 : 2);
}

// wp[py][px][m][c][u][v] = sum of w[m][c][r][s] over the taps (r,s) that fall on window position (u,v)
__global__ void up2_phase_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int MC) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;            // index into wp [4][MC][4]
  if (i >= 16 * MC) return;
  const int uv = i & 3, mc = (i >> 2) % MC, ph = i / (4 * MC);
  int r0, r1, s0, s1;
  phase_taps(ph >> 1, uv >> 1, r0, r1);
  phase_taps(ph & 1, uv & 1, s0, s1);
  const float* k = w + (size_t)mc * 9;
  float acc = 0.f;
  for (int r = r0; r <= r1; ++r)
    for (int s = s0; s <= s1; ++s) acc += k[r * 3 + s];
  wp[i] = acc;
}

// input-gradient operand: kd[c][m][dr][dc], dr/dc = 0..3 <-> dy row 2r-1+dr; taps of the 3x3 kernel per offset:
// 0:{2}  1:{1,2}  2:{0,1}  3:{0}
__global__ void up2_dgrad_weights_kernel(const float* __restrict__ w, float* __restrict__ kd, int M, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;            // index into kd [C][M][4][4]
  if (i >= 16 * M * C) return;
  const int dc = i & 3, dr = (i >> 2) & 3, m = (i >> 4) % M, c = i / (16 * M);
  const int r0 = dr == 0 ? 2 : (dr == 1 ? 1 : 0), r1 = dr == 0 ? 2 : (dr == 1 ? 2 : (dr == 2 ? 1 : 0));
  const int s0 = dc == 0 ? 2 : (dc == 1 ? 1 : 0), s1 = dc == 0 ? 2 : (dc == 1 ? 2 : (dc == 2 ? 1 : 0));
  const float* k = w + ((size_t)m * C + c) * 9;
  float acc = 0.f;
  for (int r = r0; r <= r1; ++r)
    for (int s = s0; s <= s1; ++s) acc += k[r * 3 + s];
  kd[i] = acc;
}

// dw[m][c][r][s] = sum over the (phase, window position) pairs whose tap set contains (r, s)
__global__ void up2_wgrad_combine_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int MC) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;            // index into dw [MC][3][3]
  if (i >= 9 * MC) return;
  const int s = i % 3, r = (i / 3) % 3, mc = i / 9;
  float acc = 0.f;
  for (int py = 0; py < 2; ++py)
    for (int u = 0; u < 2; ++u) {
      int r0, r1;
      phase_taps(py, u, r0, r1);
      if (r < r0 || r > r1) continue;
      for (int px = 0; px < 2; ++px)
        for (int v = 0; v < 2; ++v) {
          int s0, s1;
          phase_taps(px, v, s0, s1);
          if (s < s0 || s > s1) continue;
          acc += dwp[((size_t)(py * 2 + px) * MC + mc) * 4 + u * 2 + v];
        }
    }
  dw[i] = acc;
}

// out[ph][bc][i][j] = in[bc][2i + (ph>>1)][2j + (ph&1)]   (in: [BC][2H][2W])
__global__ __launch_bounds__(256) void space_to_depth2_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t BC, int H, int W) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // over [BC][2H][W]: one float2 of the input row
  if (i >= BC * 2 * H * W) return;
  const int j = i % W, y = (i / W) % (2 * H);
  const int64_t bc = i / ((int64_t)2 * H * W);
  const float2 v = *reinterpret_cast<const float2*>(in + (bc * 2 * H + y) * 2 * W + 2 * j);
  const int64_t plane = BC * H * W, o = (bc * H + (y >> 1)) * W + j;
  out[(size_t)((y & 1) * 2) * plane + o] = v.x;
  out[(size_t)((y & 1) * 2 + 1) * plane + o] = v.y;
}

// dx[bc][i][j] = sum of the padded-gradient entries that the replicate border maps onto (i, j)   (dp: [BC][H+2][W+2])
__global__ __launch_bounds__(256) void replicate_fold_kernel(const float* __restrict__ dp, float* __restrict__ dx, int64_t BC, int H, int W) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= BC * H * W) return;
  const int w = i % W, h = (i / W) % H;
  const int64_t bc = i / ((int64_t)H * W);
  const int Wp = W + 2;
  const float* p = dp + bc * (int64_t)(H + 2) * Wp;
  const int r0 = h == 0 ? 0 : h + 1, r1 = h == H - 1 ? H + 1 : h + 1;
  const int c0 = w == 0 ? 0 : w + 1, c1 = w == W - 1 ? W + 1 : w + 1;
  float acc = 0.f;
  for (int r = r0; r <= r1; ++r)
    for (int c = c0; c <= c1; ++c) acc += p[(int64_t)r * Wp + c];
  dx[i] = acc;
}

// The same for W % 4 == 0, four consecutive pixels per thread: an interior quad (no border row, neither border column) is ONE 16-byte load at
// p[h+1][w+1 ..] (4-byte aligned: W + 2 columns) and one aligned 16-byte store; each value is 0 + p as in the loop above.  Quads that touch the
// border run the loop per pixel.  (One pixel per thread moved 4 bytes per memory instruction and lane: 2.1 TB/s on the 8x64x240x320 gradient.)
__global__ __launch_bounds__(256) void replicate_fold4_kernel(const float* __restrict__ dp, float* __restrict__ dx, int64_t BC, int H, int W) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int W4 = W >> 2;
  if (i >= BC * H * W4) return;
  const int w = (int)(i % W4) * 4, h = (i / W4) % H;
  const int64_t bc = i / ((int64_t)H * W4);
  const int Wp = W + 2;
  const float* p = dp + bc * (int64_t)(H + 2) * Wp;
  float4 o;
  if (h != 0 && h != H - 1 && w != 0 && w != W - 4) {
    const prn_f4u q = *reinterpret_cast<const prn_f4u*>(p + (int64_t)(h + 1) * Wp + (w + 1));
    o = make_float4(0.f + q.v[0], 0.f + q.v[1], 0.f + q.v[2], 0.f + q.v[3]);
  } else {
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int wk = w + k;
      const int r0 = h == 0 ? 0 : h + 1, r1 = h == H - 1 ? H + 1 : h + 1;
      const int c0 = wk == 0 ? 0 : wk + 1, c1 = wk == W - 1 ? W + 1 : wk + 1;
      float acc = 0.f;
      for (int r = r0; r <= r1; ++r)
        for (int c = c0; c <= c1; ++c) acc += p[(int64_t)r * Wp + c];
      t[k] = acc;
    }
    o = make_float4(t[0], t[1], t[2], t[3]);
  }
  *reinterpret_cast<float4*>(dx + (bc * H + h) * (int64_t)W + w) = o;
}

}  // namespace

extern "C" int prn_up2_phase_weights(const float* w, float* wp, int M, int C, void* stream) {
  PRN_REQUIRE(w && wp && M > 0 && C > 0, "prn_up2_phase_weights: bad arguments");
  hipLaunchKernelGGL(up2_phase_weights_kernel, dim3(cdiv(16 * M * C, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, M * C);
  PRN_CHECK_LAUNCH("prn_up2_phase_weights");
  return 0;
}

extern "C" int prn_up2_dgrad_weights(const float* w, float* kd, int M, int C, void* stream) {
  PRN_REQUIRE(w && kd && M > 0 && C > 0, "prn_up2_dgrad_weights: bad arguments");
  hipLaunchKernelGGL(up2_dgrad_weights_kernel, dim3(cdiv(16 * M * C, 256)), dim3(256), 0, (hipStream_t)stream, w, kd, M, C);
  PRN_CHECK_LAUNCH("prn_up2_dgrad_weights");
  return 0;
}

extern "C" int prn_up2_wgrad_combine(const float* dwp, float* dw, int M, int C, void* stream) {
  PRN_REQUIRE(dwp && dw && M > 0 && C > 0, "prn_up2_wgrad_combine: bad arguments");
  hipLaunchKernelGGL(up2_wgrad_combine_kernel, dim3(cdiv(9 * M * C, 256)), dim3(256), 0, (hipStream_t)stream, dwp, dw, M * C);
  PRN_CHECK_LAUNCH("prn_up2_wgrad_combine");
  return 0;
}

extern "C" int prn_space_to_depth2(const float* in, float* out, int B, int C, int H, int W, void* stream) {
  PRN_REQUIRE(in && out && B > 0 && C > 0 && H > 0 && W > 0, "prn_space_to_depth2: bad arguments");
  PRN_REQUIRE((reinterpret_cast<uintptr_t>(in) & 7) == 0, "prn_space_to_depth2: input must be 8-byte aligned");
  const int64_t n = (int64_t)B * C * 2 * H * W;
  hipLaunchKernelGGL(space_to_depth2_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, (int64_t)B * C, H, W);
  PRN_CHECK_LAUNCH("prn_space_to_depth2");
  return 0;
}

extern "C" int prn_replicate_fold(const float* dp, float* dx, int B, int C, int H, int W, void* stream) {
  PRN_REQUIRE(dp && dx && B > 0 && C > 0 && H > 1 && W > 1, "prn_replicate_fold: bad arguments");
  const int64_t n = (int64_t)B * C * H * W;
  if ((W & 3) == 0 && W >= 8 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0)
    hipLaunchKernelGGL(replicate_fold4_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, dp, dx, (int64_t)B * C, H, W);
  else
    hipLaunchKernelGGL(replicate_fold_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dp, dx, (int64_t)B * C, H, W);
  PRN_CHECK_LAUNCH("prn_replicate_fold");
  return 0;
}

// Depth-error metrics of one frame (reference eval.py:164-207) as ONE fused reduction: a single pass over (pred, gt) yields
// the eight sums behind abs_rel / sq_rel / rmse / log10 / a1 / a2 / a3 (the reference runs ~25 elementwise / boolean-index
// / reduction launches).  Fixed-order fp64 partials -> deterministic.  The median ratio stays with the caller (a selection,
// not a sum).
#include "prn_common.h"

namespace {
constexpr int NQ = 8;      // count, abs_rel, sq_rel, sq_err, log10, a1, a2, a3

__global__ __launch_bounds__(256) void depth_metrics_partial_kernel(const float* __restrict__ pred, const float* __restrict__ gt, double* __restrict__ part,
                                                                    int64_t n, float dmin, float dmax) {
  double acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float g = gt[i];
    float p = pred[i];
    if (!(g > 0.5f && p > 0.5f)) continue;                  // valid_mask (eval.py:178)
    p = fminf(fmaxf(p, dmin), dmax);                        // clamp to the dataset's depth range (eval.py:189-190)
    const float d = g - p;
    const float th = fmaxf(g / p, p / g);
    acc[0] += 1.0;
    acc[1] += (double)(fabsf(d) / g);
    acc[2] += (double)(d * d / g);
    acc[3] += (double)(d * d);
    acc[4] += (double)fabsf(log10f(g) - log10f(p));
    acc[5] += th < 1.25f ? 1.0 : 0.0;
    acc[6] += th < 1.25f * 1.25f ? 1.0 : 0.0;
    acc[7] += th < 1.25f * 1.25f * 1.25f ? 1.0 : 0.0;
  }
  __shared__ double sm[4][NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double v = wave_sum_d(acc[q]);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][q] = v;
  }
  __syncthreads();
  if (threadIdx.x < NQ) part[(size_t)blockIdx.x * NQ + threadIdx.x] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

__global__ void depth_metrics_final_kernel(const double* __restrict__ part, double* __restrict__ out, int blocks) {
  const int q = threadIdx.x;
  if (q >= NQ) return;
  double t = 0.0;
  for (int b = 0; b < blocks; ++b) t += part[(size_t)b * NQ + q];
  __shared__ double s[NQ];
  s[q] = t;
  __syncthreads();
  const double cnt = s[0];
  // out: abs_rel, sq_rel, rmse, log10, a1, a2, a3, valid count
  if (q == 0) out[7] = cnt;
  else if (q == 3) out[2] = sqrt(t / cnt);
  else if (q == 1) out[0] = t / cnt;
  else if (q == 2) out[1] = t / cnt;
  else out[q - 1] = t / cnt;                                // q = 4 -> log10 (3), 5..7 -> a1..a3 (4..6)
}
}  // namespace

extern "C" int prn_depth_metrics_ws_doubles(void) { return 256 * NQ; }

extern "C" int prn_depth_metrics(const float* pred, const float* gt, double* out, double* ws, int64_t n, float min_depth, float max_depth, void* stream) {
  PRN_REQUIRE(pred && gt && out && ws && n > 0, "prn_depth_metrics: bad arguments");
  int blocks = cdiv(n, 256 * 8);
  blocks = blocks > 256 ? 256 : (blocks < 1 ? 1 : blocks);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(depth_metrics_partial_kernel, dim3(blocks), dim3(256), 0, st, pred, gt, ws, n, min_depth, max_depth);
  PRN_CHECK_LAUNCH("prn_depth_metrics/partial");
  hipLaunchKernelGGL(depth_metrics_final_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, out, blocks);
  PRN_CHECK_LAUNCH("prn_depth_metrics/final");
  return 0;
}

// Depth-error metrics of one frame (reference eval.py:164-207) as ONE fused reduction: a single pass over (pred, gt) yields
// the eight sums behind abs_rel / sq_rel / rmse / log10 / a1 / a2 / a3 (the reference runs ~25 elementwise / boolean-index
// / reduction launches).  Fixed-order fp64 partials -> deterministic.  The median ratio stays with the caller (a selection,
// not a sum).
//
// Pairwise IoU of one frame's detections against its ground truth (reference eval.py:214-215 -> funcs.py:30-71): the
// reference multiplies the two [n, H*W] float mask matrices; masks are binary, so the intersection of a pair is a population
// count.  Here every mask is packed to one bit per pixel once (H*W/8 bytes, read back from L2 by every pair) and a pair costs
// H*W/32 AND+popcount steps -- integer-exact, so the fp32 quotient equals the reference's bit for bit.
#include "prn_common.h"

namespace {
// bit i of word w of mask m = (mask[m][32 w + i] != 0); area[m] += popcount (integer atomics: order-independent)
__global__ __launch_bounds__(256) void mask_pack_kernel(const unsigned char* __restrict__ masks, unsigned* __restrict__ bits, unsigned* __restrict__ area,
                                                        int64_t HW, int words) {
  const int m = blockIdx.y;
  const int w = blockIdx.x * 256 + threadIdx.x;
  unsigned word = 0;
  if (w < words) {
    const unsigned char* src = masks + (size_t)m * HW + (size_t)w * 32;
    const int64_t left = HW - (int64_t)w * 32;
    if (left >= 32 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
      const uint4 q0 = reinterpret_cast<const uint4*>(src)[0], q1 = reinterpret_cast<const uint4*>(src)[1];
      const unsigned v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int b = 0; b < 4; ++b) word |= ((v[k] >> (8 * b)) & 0xffu) ? (1u << (4 * k + b)) : 0u;
    } else {
      for (int i = 0; i < 32 && i < left; ++i) word |= src[i] ? (1u << i) : 0u;
    }
    bits[(size_t)m * words + w] = word;
  }
  int cnt = __popc(word);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(area + m, (unsigned)cnt);
}

// one workgroup per (a, b) pair: intersection = sum_w popcount(bits_a[w] & bits_b[w]); IoU in fp32 with the reference's
// operation order (funcs.py:67-71: intersection / (area_a + area_b - intersection); 0/0 stays NaN).  Thread 0 also writes
// the pair's box IoU (funcs.py:23-55), every product / sum rounded separately like the tensor ops (fp contract off).
__global__ __launch_bounds__(256) void pair_iou_kernel(const unsigned* __restrict__ bits_a, const unsigned* __restrict__ bits_b,
                                                       const unsigned* __restrict__ area_a, const unsigned* __restrict__ area_b,
                                                       const float* __restrict__ box_a, const float* __restrict__ box_b, float* __restrict__ mask_iou,
                                                       float* __restrict__ box_iou, int B, int words) {
  const int a = blockIdx.y, b = blockIdx.x;
  if (bits_a) {
    const unsigned* pa = bits_a + (size_t)a * words;
    const unsigned* pb = bits_b + (size_t)b * words;
    int cnt = 0;
    for (int w = threadIdx.x; w < words; w += 256) cnt += __popc(pa[w] & pb[w]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    __shared__ int sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float inter = (float)(sm[0] + sm[1] + sm[2] + sm[3]);
      mask_iou[(size_t)a * B + b] = inter / (((float)area_a[a] + (float)area_b[b]) - inter);      // (integers < 2^24: exact sums)
    }
  }
  if (box_a && threadIdx.x == 0) {
#pragma clang fp contract(off)      // every product / sum rounded on its own (contracted, area_a + w_b * h_b became one fma: 1 ulp off in 6 % of the pairs)
    const float* p = box_a + 4 * a;
    const float* q = box_b + 4 * b;
    const float w = fmaxf(fminf(p[2], q[2]) - fmaxf(p[0], q[0]), 0.f);
    const float h = fmaxf(fminf(p[3], q[3]) - fmaxf(p[1], q[1]), 0.f);
    const float inter = w * h;
    const float aa = (p[2] - p[0]) * (p[3] - p[1]);
    const float ab = (q[2] - q[0]) * (q[3] - q[1]);
    const float uni = aa + ab;
    box_iou[(size_t)a * B + b] = inter / (uni - inter);
  }
}

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

// ---- pairwise IoU -------------------------------------------------------------------------------------------------
namespace {
inline size_t iou_words(int64_t HW) { return (size_t)((HW + 31) / 32); }
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
// Category scores of one grid level for the post-process: sigmoid, then the 2x2 "point NMS" of models/functions/nms.py:8-12 (a cell
// keeps its score iff it is the maximum of the window {i-1, i} x {j-1, j}), written channels-last into the level's rows of the
// [B, cells, C] score matrix the candidate selection reads -- sigmoid + max_pool2d + eq + float + mul + permute + cat in one launch.
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void sigmoid_point_nms_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int S, int64_t out_bs, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // over (b, c, y, x) of the input
  if (i >= total) return;
  const int xx = i % S, yy = (i / S) % S, c = (i / ((int64_t)S * S)) % C;
  const int64_t b = i / ((int64_t)S * S * C);
  const float* plane = x + (b * C + c) * (int64_t)S * S;
  auto sig = [](float v) { return 1.f / (1.f + expf(-v)); };
  const float h = sig(plane[yy * S + xx]);
  float m = h;
  if (xx > 0) m = fmaxf(m, sig(plane[yy * S + xx - 1]));
  if (yy > 0) m = fmaxf(m, sig(plane[(yy - 1) * S + xx]));
  if (xx > 0 && yy > 0) m = fmaxf(m, sig(plane[(yy - 1) * S + xx - 1]));
  out[b * out_bs + ((int64_t)yy * S + xx) * C + c] = (m == h) ? h : 0.f;
}
#pragma clang fp contract(on)

// Per soft mask (row of n x HW sigmoid values): number of values above the threshold and their sum -- the reference's
// seg_masks.sum((1, 2)) and (seg_preds * seg_masks.float()).sum((1, 2)) (planerecnet.py:227-240) in one pass.  One workgroup per
// row, fixed summation order (strided per-thread partials, then a tree over the 256 partials): the result of a row does not depend
// on how many rows the launch has, i.e. on which other images share the batch.
__global__ __launch_bounds__(256) void mask_stats_kernel(const float* __restrict__ seg, float* __restrict__ count, float* __restrict__ msum, int64_t HW, float thr) {
  const float* row = seg + (size_t)blockIdx.x * HW;
  float c = 0.f, s = 0.f;
  for (int64_t i = threadIdx.x; i < HW; i += 256) {
    const float v = row[i];
    if (v > thr) { c += 1.f; s += v; }
  }
  __shared__ float sc[256], ss[256];
  sc[threadIdx.x] = c; ss[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { sc[threadIdx.x] += sc[threadIdx.x + o]; ss[threadIdx.x] += ss[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { count[blockIdx.x] = sc[0]; msum[blockIdx.x] = ss[0]; }
}

// Tight boxes of n binary masks (planerecnet.py:282-287: per instance torch.where -> min / max of the set rows and columns).  One
// workgroup per mask (a frame has a handful of detections: 1024 threads reading 16-byte words, so that one workgroup streams its
// 300 KB mask in ~20 loads per thread); a mask without a set pixel gets (H + W, H + W, -1, -1) like the vectorised torch form.
__global__ __launch_bounds__(1024) void mask_boxes_kernel(const unsigned char* __restrict__ masks, float* __restrict__ boxes, int H, int W) {
  const unsigned char* m = masks + (size_t)blockIdx.x * H * W;
  int x0 = H + W, y0 = H + W, x1 = -1, y1 = -1;
  if ((W & 15) == 0 && (reinterpret_cast<uintptr_t>(m) & 15) == 0) {          // a 16-byte word never straddles two rows
    const int words = (H * W) >> 4, wpr = W >> 4;
    const uint4* q = reinterpret_cast<const uint4*>(m);
    for (int i = threadIdx.x; i < words; i += 1024) {
      const uint4 v = q[i];
      if ((v.x | v.y | v.z | v.w) == 0u) continue;
      const int y = i / wpr, xb = (i - y * wpr) << 4;
      const unsigned w4[4] = {v.x, v.y, v.z, v.w};
      int lo = 16, hi = -1;
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if ((w4[k] >> (8 * t)) & 0xffu) { lo = min(lo, 4 * k + t); hi = max(hi, 4 * k + t); }
      x0 = min(x0, xb + lo); x1 = max(x1, xb + hi);
      y0 = min(y0, y); y1 = max(y1, y);
    }
  } else {
    for (int y = threadIdx.x >> 6; y < H; y += 16) {          // a wave per row: coalesced byte reads
      const unsigned char* row = m + (size_t)y * W;
      bool any = false;
      for (int x = threadIdx.x & 63; x < W; x += 64)
        if (row[x]) { any = true; x0 = min(x0, x); x1 = max(x1, x); }
      if (any) { y0 = min(y0, y); y1 = max(y1, y); }
    }
  }
  __shared__ int sm[4];
  if (threadIdx.x == 0) { sm[0] = H + W; sm[1] = H + W; sm[2] = -1; sm[3] = -1; }
  __syncthreads();
  if (x1 >= 0) { atomicMin(&sm[0], x0); atomicMin(&sm[1], y0); atomicMax(&sm[2], x1); atomicMax(&sm[3], y1); }      // (integers: order-free)
  __syncthreads();
  if (threadIdx.x < 4) boxes[(size_t)blockIdx.x * 4 + threadIdx.x] = (float)sm[threadIdx.x];
}

// Matrix NMS (models/functions/nms.py:15-50) on the [n, n] mask-IoU matrix of detections sorted by score: for detection j,
//   decay[i][j] = iou[i][j] if i < j and label_i == label_j else 0;  comp_i = max_k decay[k][i];
//   coef_j = min_i  exp(-sigma decay[i][j]^2) / exp(-sigma comp_i^2)        (gaussian)   |   (1 - decay[i][j]) / (1 - comp_i)   (linear)
// over ALL i (rows at or below the diagonal contribute 1 / exp(-sigma comp_i^2), as in the dense torch form).  Two launches (the
// second needs every comp_i): one WAVE per column j, its lanes stride over the rows, max / min across the wave -- order-free, so the
// result equals the dense form bit for bit.
#pragma clang fp contract(off)
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

__global__ __launch_bounds__(256) void matrix_nms_den_kernel(const float* __restrict__ iou, const int64_t* __restrict__ labels, float* __restrict__ den,
                                                             int n, float sigma, int gaussian) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= n) return;
  const int64_t lj = labels[j];
  float comp = 0.f;                                               // (column maximum of a matrix whose diagonal and lower part are 0)
  for (int i = lane; i < j; i += 64) comp = fmaxf(comp, labels[i] == lj ? iou[(size_t)i * n + j] : 0.f);
  comp = wave_max(comp);
  if (lane == 0) den[j] = gaussian ? expf(-sigma * (comp * comp)) : 1.f - comp;
}

__global__ __launch_bounds__(256) void matrix_nms_coef_kernel(const float* __restrict__ iou, const int64_t* __restrict__ labels, const float* __restrict__ scores,
                                                              const float* __restrict__ den, float* __restrict__ out, int n, float sigma, int gaussian) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= n) return;
  const int64_t lj = labels[j];
  float coef = INFINITY;
  for (int i = lane; i < n; i += 64) {
    const float d = (i < j && labels[i] == lj) ? iou[(size_t)i * n + j] : 0.f;
    const float num = gaussian ? expf(-sigma * (d * d)) : 1.f - d;
    coef = fminf(coef, num / den[i]);
  }
  coef = wave_min(coef);
  if (lane == 0) out[j] = scores[j] * coef;
}
#pragma clang fp contract(on)

}  // namespace

extern "C" int64_t prn_pairwise_iou_ws_bytes(int A, int B, int64_t HW) {
  if (A <= 0 || B <= 0 || HW <= 0) return 256;
  return (int64_t)(align256((size_t)(A + B) * 4) + (size_t)(A + B) * iou_words(HW) * 4);
}

extern "C" int prn_pairwise_iou(const unsigned char* masks_a, const unsigned char* masks_b, const float* boxes_a, const float* boxes_b, int A, int B,
                                int64_t HW, float* mask_iou, float* box_iou, void* ws, void* stream) {
  PRN_REQUIRE(A > 0 && B > 0 && A < 65536 && B < 65536, "prn_pairwise_iou: needs 1 <= A, B < 65536 (got %d, %d)", A, B);
  const bool want_mask = masks_a != nullptr, want_box = boxes_a != nullptr;
  PRN_REQUIRE(want_mask || want_box, "prn_pairwise_iou: neither masks nor boxes given");
  PRN_REQUIRE(!want_mask || (masks_b && mask_iou && ws && HW > 0 && HW < (1LL << 24)),
              "prn_pairwise_iou: masks need masks_b, mask_iou, ws and 0 < H*W < 2^24 (fp32-exact areas)");
  PRN_REQUIRE(!want_box || (boxes_b && box_iou), "prn_pairwise_iou: boxes need boxes_b and box_iou");
  hipStream_t st = (hipStream_t)stream;
  unsigned *area = nullptr, *bits = nullptr;
  const int words = (int)iou_words(HW > 0 ? HW : 1);
  if (want_mask) {
    area = (unsigned*)ws;
    bits = (unsigned*)((char*)ws + align256((size_t)(A + B) * 4));
    PRN_REQUIRE(hipMemsetAsync(area, 0, (size_t)(A + B) * 4, st) == hipSuccess, "prn_pairwise_iou: memset failed");
    hipLaunchKernelGGL(mask_pack_kernel, dim3(cdiv(words, 256), A), dim3(256), 0, st, masks_a, bits, area, HW, words);
    hipLaunchKernelGGL(mask_pack_kernel, dim3(cdiv(words, 256), B), dim3(256), 0, st, masks_b, bits + (size_t)A * words, area + A, HW, words);
    PRN_CHECK_LAUNCH("prn_pairwise_iou/pack");
  }
  hipLaunchKernelGGL(pair_iou_kernel, dim3(B, A), dim3(256), 0, st, (const unsigned*)bits, (const unsigned*)(bits ? bits + (size_t)A * words : nullptr),
                     (const unsigned*)area, (const unsigned*)(area ? area + A : nullptr), want_box ? boxes_a : nullptr, boxes_b, mask_iou, box_iou, B, words);
  PRN_CHECK_LAUNCH("prn_pairwise_iou/pairs");
  return 0;
}

extern "C" int prn_mask_boxes(const unsigned char* masks, int n, int H, int W, float* boxes, void* stream) {
  PRN_REQUIRE(masks && boxes && n > 0 && H > 0 && W > 0, "prn_mask_boxes: bad arguments");
  hipLaunchKernelGGL(mask_boxes_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, masks, boxes, H, W);
  PRN_CHECK_LAUNCH("prn_mask_boxes");
  return 0;
}

extern "C" int prn_matrix_nms(const float* iou, const int64_t* labels, const float* scores, int n, float sigma, int gaussian, float* out, float* ws,
                              void* stream) {
  PRN_REQUIRE(iou && labels && scores && out && ws && n > 0, "prn_matrix_nms: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(matrix_nms_den_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, iou, labels, ws, n, sigma, gaussian);
  PRN_CHECK_LAUNCH("prn_matrix_nms/den");
  hipLaunchKernelGGL(matrix_nms_coef_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, iou, labels, scores, (const float*)ws, out, n, sigma, gaussian);
  PRN_CHECK_LAUNCH("prn_matrix_nms/coef");
  return 0;
}

extern "C" int prn_mask_stats(const float* seg, int n, int64_t HW, float thr, float* count, float* msum, void* stream) {
  PRN_REQUIRE(seg && count && msum && n > 0 && HW > 0 && HW < (1LL << 24), "prn_mask_stats: bad arguments (0 < H*W < 2^24: counts are exact floats)");
  hipLaunchKernelGGL(mask_stats_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, seg, count, msum, HW, thr);
  PRN_CHECK_LAUNCH("prn_mask_stats");
  return 0;
}

extern "C" int prn_sigmoid_point_nms(const float* x, float* out, int B, int C, int S, int64_t out_batch_stride, void* stream) {
  PRN_REQUIRE(x && out && B > 0 && C > 0 && S > 0 && out_batch_stride >= (int64_t)S * S * C, "prn_sigmoid_point_nms: bad arguments");
  const int64_t total = (int64_t)B * C * S * S;
  hipLaunchKernelGGL(sigmoid_point_nms_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, out, C, S, out_batch_stride, total);
  PRN_CHECK_LAUNCH("prn_sigmoid_point_nms");
  return 0;
}

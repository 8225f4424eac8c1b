// GT-only preparation of the joint loss on the device: the per-instance mask statistics and the 1/4-scale masks
// of SOLOv2's target assignment (reference models/functions/losses.py:200-286) and the virtual-normal triplet sampling
// (models/functions/vnl.py:43-70, 119-140), which the reference -- and rounds 1-2 of this build, in worker processes -- run on the host
// per image and per plane.  HBM-bound byte work: nothing here is a GEMM.
//
// A REGION is a set of pixels of one image in raster order: the N_b plane masks of image b (regions img_first[b] .. img_first[b+1]-1
// of the packed mask tensor) and, per image, the pixels no plane covers (region Ntot + b).  Every region is cut into 64-pixel
// segments; prn_gt_mask_stats counts the set pixels of each segment (one 64-byte read per lane) and accumulates the region's
// totals -- pixel count, sum of x, sum of y: exact integers, so the atomics are order-independent; prn_gt_segment_scan turns the
// counts into an exclusive prefix per region; prn_gt_sample_triplets then maps a RANK r (the r-th set pixel of the region in
// raster order -- what np.flatnonzero(mask)[r] is on the host) to its pixel by a binary search over the prefix and a scan of one
// 64-byte segment.  Ranks are either injected (the reference's numpy stream drawn on the host: bit-identical triplets, used by the
// parity tests) or drawn here by a counter-based generator (Philox4x32-10 keyed by (seed, sample index): production).
#include "prn_common.h"

namespace {

constexpr int SEG = 64;                       // pixels per segment

__device__ __forceinline__ uint4 load16(const unsigned char* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ unsigned nz_bytes(unsigned v) {          // 0x01 in every byte of v that is non-zero
  v |= v >> 4; v |= v >> 2; v |= v >> 1;
  return v & 0x01010101u;
}

// bit k of the result = pixel k of the 64-pixel segment `s` of region `r` is set
__device__ __forceinline__ unsigned long long segment_bits(const unsigned char* __restrict__ masks, const int* __restrict__ img_first, int Ntot, int r, int s,
                                                           long long HW) {
  unsigned long long bits = 0;
  auto gather = [&](const unsigned char* base) {
    unsigned long long b = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 v = load16(base + q * 16);
      const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned n = nz_bytes(w[j]);                      // bytes -> bits 0, 8, 16, 24
        const unsigned four = (n & 1u) | ((n >> 7) & 2u) | ((n >> 14) & 4u) | ((n >> 21) & 8u);
        b |= (unsigned long long)four << (q * 16 + j * 4);
      }
    }
    return b;
  };
  if (r < Ntot) return gather(masks + (size_t)r * HW + (size_t)s * SEG);
  const int b = r - Ntot;                                       // the pixels of image b that no plane covers
  for (int i = img_first[b]; i < img_first[b + 1]; ++i) bits |= gather(masks + (size_t)i * HW + (size_t)s * SEG);
  return ~bits;
}

// grid (ceil(nseg / 256), R): one lane per segment.  segcnt [R][nseg] uint8, totals [R][3] (count, sum x, sum y) zeroed by the caller.
__global__ __launch_bounds__(256) void gt_mask_stats_kernel(const unsigned char* __restrict__ masks, const int* __restrict__ img_first, int Ntot, int W,
                                                            int nseg, long long HW, unsigned char* __restrict__ segcnt,
                                                            unsigned long long* __restrict__ totals) {
  const int r = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
  unsigned long long cnt = 0, sx = 0, sy = 0;
  if (s < nseg) {
    const unsigned long long bits = segment_bits(masks, img_first, Ntot, r, s, HW);
    cnt = __popcll(bits);
    segcnt[(size_t)r * nseg + s] = (unsigned char)cnt;
    // a segment lies inside one image row when W % 64 == 0 (checked by the host); otherwise per pixel
    const int p0 = s * SEG;
    if (W % SEG == 0) {
      const int y = p0 / W, x0 = p0 - y * W;
      unsigned long long b = bits;
      unsigned xs = 0;
      while (b) { const int k = __ffsll((long long)b) - 1; xs += k; b &= b - 1; }
      sx = (unsigned long long)x0 * cnt + xs;
      sy = (unsigned long long)y * cnt;
    } else {
      unsigned long long b = bits;
      while (b) { const int k = __ffsll((long long)b) - 1; const int p = p0 + k; sx += p % W; sy += p / W; b &= b - 1; }
    }
  }
  // workgroup reduction, then three atomics per workgroup (exact integer sums: the order does not matter)
  __shared__ unsigned long long sm[3][4];
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64); sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64);
  }
  if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = cnt; sm[1][threadIdx.x >> 6] = sx; sm[2][threadIdx.x >> 6] = sy; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const unsigned long long t = sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3];
    if (t) atomicAdd(totals + (size_t)r * 3 + threadIdx.x, t);
  }
}

// one workgroup per region: segstart[r][0 .. nseg] = exclusive prefix of segcnt[r][.]
__global__ __launch_bounds__(256) void gt_segment_scan_kernel(const unsigned char* __restrict__ segcnt, int nseg, int* __restrict__ segstart) {
  const int r = blockIdx.x;
  const unsigned char* c = segcnt + (size_t)r * nseg;
  int* out = segstart + (size_t)r * (nseg + 1);
  const int per = (nseg + 255) / 256, beg = threadIdx.x * per, end = min(beg + per, nseg);
  int sum = 0;
  for (int i = beg; i < end; ++i) sum += c[i];
  __shared__ int sm[256];
  sm[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {                           // inclusive scan of the 256 partial sums
    const int v = threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
    __syncthreads();
    sm[threadIdx.x] += v;
    __syncthreads();
  }
  int run = sm[threadIdx.x] - sum;
  for (int i = beg; i < end; ++i) { out[i] = run; run += c[i]; }
  if (threadIdx.x == 255) out[nseg] = sm[255];
}

// OpenCV INTER_LINEAR at exactly 1/4: the mean of the 2x2 centre pixels of every 4x4 block, rounded half up (the reference's
// imrescale(mask, 0.25), losses.py:243-247; same closed form as funcs.quarter_mask_u8).  One lane per output pixel.
__global__ __launch_bounds__(256) void gt_quarter_masks_kernel(const unsigned char* __restrict__ masks, unsigned char* __restrict__ out, long long total, int H,
                                                               int W) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int w4 = W / 4, h4 = H / 4;
  const int x = (int)(i % w4), y = (int)((i / w4) % h4);
  const long long n = i / ((long long)w4 * h4);
  const unsigned char* p = masks + (size_t)n * H * W + (size_t)(4 * y + 1) * W + 4 * x + 1;
  const int s = p[0] + p[1] + p[W] + p[W + 1];
  out[i] = (unsigned char)((s + 2) >> 2);
}

// Philox4x32-10 (Salmon et al., SC'11): counter (i lo, i hi, call lo, call hi), key (seed lo, seed hi) -> four 32-bit words
__device__ __forceinline__ void philox(unsigned long long idx, unsigned long long call, unsigned long long seed, unsigned out[4]) {
  unsigned c0 = (unsigned)idx, c1 = (unsigned)(idx >> 32), c2 = (unsigned)call, c3 = (unsigned)(call >> 32), k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// One lane per sampled triplet t of the batch (segment g = triplet's sampled region, listed in seg_region / seg_first): the three
// points' pixel ids.  ranks != NULL: the three ranks of triplet t are ranks[j * n_tot + t] (injected: the host's numpy stream);
// NULL: drawn here, rank = floor(u * count) with u uniform in [0, 1).  gid [3][n_tot] = image * HW + pixel.
__global__ __launch_bounds__(256) void gt_sample_triplets_kernel(const unsigned char* __restrict__ masks, const int* __restrict__ img_first, int Ntot, int nseg,
                                                                 long long HW, const int* __restrict__ segstart, const int* __restrict__ trip_seg,
                                                                 const int* __restrict__ seg_region, const int* __restrict__ seg_img,
                                                                 const int* __restrict__ ranks, unsigned long long seed, unsigned long long call,
                                                                 long long n_tot, int* __restrict__ gid) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_tot) return;
  const int g = trip_seg[t], r = seg_region[g];
  const int* st = segstart + (size_t)r * (nseg + 1);
  const int count = st[nseg];
  unsigned rnd[4];
  if (!ranks) philox((unsigned long long)t, call, seed, rnd);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    int rank;
    if (ranks) rank = ranks[(size_t)j * n_tot + t];
    else { rank = (int)(((unsigned long long)rnd[j] * (unsigned long long)count) >> 32); }      // floor(u * count), u = rnd / 2^32
    // last segment whose first rank is <= rank
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (st[mid] <= rank) lo = mid; else hi = mid - 1;
    }
    unsigned long long bits = segment_bits(masks, img_first, Ntot, r, lo, HW);
    for (int k = rank - st[lo]; k > 0; --k) bits &= bits - 1;                                   // drop the k lower set bits
    const int pos = __ffsll((long long)bits) - 1;
    gid[(size_t)j * n_tot + t] = seg_img[g] * (int)HW + lo * SEG + pos;
  }
}

}  // namespace

extern "C" int64_t prn_gt_segments(int H, int W) { return ((int64_t)H * W) / SEG; }

extern "C" int prn_gt_mask_stats(const unsigned char* masks, const int* img_first, int B, int Ntot, int H, int W, unsigned char* segcnt, int* segstart,
                                 unsigned long long* totals, void* stream) {
  PRN_REQUIRE(masks && img_first && segcnt && segstart && totals && B > 0 && Ntot >= 0 && H > 0 && W > 0, "prn_gt_mask_stats: bad arguments");
  PRN_REQUIRE(((int64_t)H * W) % SEG == 0 && (reinterpret_cast<uintptr_t>(masks) & 15) == 0, "prn_gt_mask_stats: H * W must be a multiple of 64, masks 16-byte aligned");
  const int nseg = (int)(((int64_t)H * W) / SEG), R = Ntot + B;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(totals, 0, (size_t)R * 3 * sizeof(unsigned long long), st) != hipSuccess) { prn_set_error("prn_gt_mask_stats: memset failed"); return 1; }
  hipLaunchKernelGGL(gt_mask_stats_kernel, dim3(cdiv(nseg, 256), R), dim3(256), 0, st, masks, img_first, Ntot, W, nseg, (long long)H * W, segcnt, totals);
  PRN_CHECK_LAUNCH("prn_gt_mask_stats");
  hipLaunchKernelGGL(gt_segment_scan_kernel, dim3(R), dim3(256), 0, st, (const unsigned char*)segcnt, nseg, segstart);
  PRN_CHECK_LAUNCH("prn_gt_mask_stats/scan");
  return 0;
}

extern "C" int prn_gt_quarter_masks(const unsigned char* masks, unsigned char* out, int N, int H, int W, void* stream) {
  PRN_REQUIRE(N >= 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0, "prn_gt_quarter_masks: bad arguments (H, W multiples of 4)");
  const long long total = (long long)N * (H / 4) * (W / 4);
  if (total == 0) return 0;                                    // (a batch without a single plane: nothing to do, the pointers may be NULL)
  PRN_REQUIRE(masks && out, "prn_gt_quarter_masks: null tensor");
  hipLaunchKernelGGL(gt_quarter_masks_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, masks, out, total, H, W);
  PRN_CHECK_LAUNCH("prn_gt_quarter_masks");
  return 0;
}

extern "C" int prn_gt_sample_triplets(const unsigned char* masks, const int* img_first, int B, int Ntot, int H, int W, const int* segstart, const int* trip_seg,
                                      const int* seg_region, const int* seg_img, const int* ranks, unsigned long long seed, unsigned long long call, int64_t n_tot, int* gid,
                                      void* stream) {
  PRN_REQUIRE(masks && img_first && segstart && trip_seg && seg_region && seg_img && gid && B > 0 && n_tot >= 0, "prn_gt_sample_triplets: bad arguments");
  PRN_REQUIRE((int64_t)B * H * W < (1LL << 31), "prn_gt_sample_triplets: batch too large for 32-bit pixel ids");
  if (n_tot == 0) return 0;
  const int nseg = (int)(((int64_t)H * W) / SEG);
  hipLaunchKernelGGL(gt_sample_triplets_kernel, dim3(cdiv(n_tot, 256)), dim3(256), 0, (hipStream_t)stream, masks, img_first, Ntot, nseg, (long long)H * W, segstart,
                     trip_seg, seg_region, seg_img, ranks, seed, call, (long long)n_tot, gid);
  PRN_CHECK_LAUNCH("prn_gt_sample_triplets");
  return 0;
}

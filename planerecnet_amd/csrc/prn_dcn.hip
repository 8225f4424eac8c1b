// DCNv2 sampling: offset gather + bilinear sample + modulation, forward and backward.
// One thread owns one (b, tap, ho, wo): the sampling position, the four corner weights and the corner
// validity are computed once and reused for every input channel; the channel loop streams x planes with the
// wave's 64 consecutive wo positions touching neighbouring addresses, and writes the column tensor fully
// coalesced.  HBM-bound (no contraction here; the [Cout x 9Cin] GEMM runs on the MFMA conv kernel).
#include <stdlib.h>
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

// Where the offsets / modulation come from.  raw = 1: `off` is the raw [B,27,Ho,Wo] output of the merged offset|modulator
// conv (offsets clamped to +-maxoff, modulation 2*sigmoid: models/dcn.py:53-57 folded in).  raw = 0: torchvision semantics,
// `off` [B,18,Ho,Wo] and `msk` [B,9,Ho,Wo] (or NULL = 1.0) are used as given.
struct OmView {
  const float* off; const float* msk;
  int64_t off_bs, msk_bs;      // elements per image
  float maxoff;
  int raw, pad;
};

// Sampling geometry of torchvision's deform_conv2d (see oracle/dcn_ref.py for the restated rule).
__device__ __forceinline__ Tap make_tap(const OmView& v, int b, int k, int ho, int wo, int Ho, int Wo,
                                        int H, int W, int stride, bool need_grad) {
  Tap t;
  const size_t plane = (size_t)Ho * Wo, pix = (size_t)ho * Wo + wo;
  const float* ob = v.off + (size_t)b * v.off_bs;
  const float ry = ob[(2 * k) * plane + pix], rx = ob[(2 * k + 1) * plane + pix];
  float dy = ry, dx = rx;
  t.pass_y = t.pass_x = true;
  if (v.raw) {
    const float maxoff = v.maxoff;
    const float rm = ob[(18 + k) * plane + pix];
    dy = fminf(fmaxf(ry, -maxoff), maxoff); dx = fminf(fmaxf(rx, -maxoff), maxoff);
    t.pass_y = (ry >= -maxoff) && (ry <= maxoff);
    t.pass_x = (rx >= -maxoff) && (rx <= maxoff);
    t.sig = 1.f / (1.f + expf(-rm));
    t.mod = 2.f * t.sig;
  } else {
    t.sig = 0.f;
    t.mod = v.msk ? v.msk[(size_t)b * v.msk_bs + (size_t)k * plane + pix] : 1.f;
  }
  const int ki = k / 3, kj = k - ki * 3;
  const float y = (float)(ho * stride - v.pad + ki) + dy, x = (float)(wo * stride - v.pad + kj) + dx;
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
__global__ __launch_bounds__(256) void dcn_sample_kernel(const float* __restrict__ x, OmView om,
                                                         float* __restrict__ cols, int B, int C, int H, int W, int Ho,
                                                         int Wo, int stride) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t plane = (int64_t)Ho * Wo;
  if (gid >= (int64_t)B * 9 * plane) return;
  const int pix = gid % plane, k = (gid / plane) % 9, b = gid / (9 * plane);
  const int ho = pix / Wo, wo = pix - ho * Wo;
  const Tap t = make_tap(om, b, k, ho, wo, Ho, Wo, H, W, stride, false);
  const int cper = (C + gridDim.y - 1) / gridDim.y;
  const int c0 = blockIdx.y * cper, c1 = min(C, c0 + cper);
  const size_t HW = (size_t)H * W;
  const float* xp = x + ((size_t)b * C + c0) * HW;
  float* cp = cols + (((size_t)b * C + c0) * 9 + k) * plane + pix;
  // Loads of UN channels are issued together, then consumed: written as a plain unrolled loop hipcc emitted one
  // global_load + s_waitcnt vmcnt(0) per corner (one memory round trip per FMA); sched_barrier keeps the two groups apart.
  constexpr int UN = 8;
  int c = c0;
  for (; c + UN <= c1; c += UN) {
    float a00[UN], a01[UN], a10[UN], a11[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const float* q = xp + (size_t)u * HW;
      a00[u] = q[t.i00]; a01[u] = q[t.i01]; a10[u] = q[t.i10]; a11[u] = q[t.i11];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < UN; ++u) cp[(size_t)u * 9 * plane] = t.mod * (t.w00 * a00[u] + t.w01 * a01[u] + t.w10 * a10[u] + t.w11 * a11[u]);
    xp += (size_t)UN * HW;
    cp += (size_t)UN * 9 * plane;
  }
  for (; c < c1; ++c) {
    const float v = t.w00 * xp[t.i00] + t.w01 * xp[t.i01] + t.w10 * xp[t.i10] + t.w11 * xp[t.i11];
    *cp = t.mod * v;
    xp += HW;
    cp += 9 * plane;
  }
}

// backward, part 1: gradients of the raw offset / modulator maps. Thread per (b, tap, pixel), channel loop split over
// blockIdx.y into fixed-order partials (no atomics): part[g][b][27][Ho*Wo].
__global__ __launch_bounds__(256) void dcn_dom_partial_kernel(const float* __restrict__ x, OmView om,
                                                              const float* __restrict__ dcols, float* __restrict__ part, int B,
                                                              int C, int H, int W, int Ho, int Wo, int stride) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t plane = (int64_t)Ho * Wo;
  if (gid >= (int64_t)B * 9 * plane) return;
  const int pix = gid % plane, k = (gid / plane) % 9, b = gid / (9 * plane);
  const int ho = pix / Wo, wo = pix - ho * Wo;
  const Tap t = make_tap(om, b, k, ho, wo, Ho, Wo, H, W, stride, true);
  const int cper = (C + gridDim.y - 1) / gridDim.y;
  const int c0 = blockIdx.y * cper, c1 = min(C, c0 + cper);
  const size_t HW = (size_t)H * W;
  const float* xp = x + ((size_t)b * C + c0) * HW;
  const float* dp = dcols + (((size_t)b * C + c0) * 9 + k) * plane + pix;
  float gy = 0.f, gx = 0.f, gm = 0.f;
  constexpr int UN = 8;                                    // (same load grouping as dcn_sample_kernel)
  int c = c0;
  for (; c + UN <= c1; c += UN) {
    float g[UN], a00[UN], a01[UN], a10[UN], a11[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const float* q = xp + (size_t)u * HW;
      g[u] = dp[(size_t)u * 9 * plane];
      a00[u] = q[t.i00]; a01[u] = q[t.i01]; a10[u] = q[t.i10]; a11[u] = q[t.i11];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      gm += g[u] * (t.w00 * a00[u] + t.w01 * a01[u] + t.w10 * a10[u] + t.w11 * a11[u]);
      gy += g[u] * (t.gy00 * a00[u] + t.gy01 * a01[u] + t.gy10 * a10[u] + t.gy11 * a11[u]);
      gx += g[u] * (t.gx00 * a00[u] + t.gx01 * a01[u] + t.gx10 * a10[u] + t.gx11 * a11[u]);
    }
    xp += (size_t)UN * HW;
    dp += (size_t)UN * 9 * plane;
  }
  for (; c < c1; ++c) {
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
  ob[(18 + k) * plane] = om.raw ? gm * 2.f * t.sig * (1.f - t.sig) : gm;      // (raw: through 2*sigmoid; else d-mask itself)
}

// sums the channel-group partials [G][B][27][plane]; d_msk == NULL: one [B,27,plane] tensor, else d_off [B,18,plane] + d_msk [B,9,plane]
__global__ void dcn_dom_final_kernel(const float* __restrict__ part, float* __restrict__ d_off, float* __restrict__ d_msk, int64_t n, int G,
                                     int plane) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int g = 0; g < G; ++g) s += part[(size_t)g * n + i];
  if (d_msk == nullptr) { d_off[i] = s; return; }
  const int64_t b = i / (27 * (int64_t)plane), r = i - b * 27 * (int64_t)plane;
  if (r < 18 * (int64_t)plane) d_off[b * 18 * plane + r] = s;
  else d_msk[b * 9 * plane + (r - 18 * (int64_t)plane)] = s;
}

// backward, part 2: d-input as a GATHER.  Every sampling point (b, tap k, output pixel p) spreads its column gradient onto
// up to four input pixels with weights that do not depend on the channel, so the scatter pattern is inverted once per call
// into a CSR structure -- bins (b, k, q) over input pixels q, entries (source column index k*P + p, modulator * corner
// weight) -- and dx is then accumulated in registers, channel by channel, by a thread that owns (b, q, CH channels):
// no atomics on floats, no LDS-sized limit on the plane, dx written exactly once with coalesced stores.  (The first
// version privatised the scatter in LDS; ds_add_f32 retires about one lane per clock per CU, which made that kernel
// 415 us on the 30x40 stage -- more than the GEMMs of the whole block.)  Binning by tap and walking the taps in lock
// step keeps a wave's gathers inside one column plane and within a few pixels of each other (two or three cache lines);
// a pixel-major bin order with one flat entry loop per pixel was measured 40 % slower for exactly that reason.
// Entry order inside a bin follows the atomic cursor, i.e. the fp32 summation order is not fixed (neither is
// torchvision's atomicAdd backward); d_om stays deterministic.
struct CsrEntry { int src; float w; };

// Fixed order inside a bin (by source point, then weight bits).  The fills hand out a bin's slots with atomics, i.e. in arrival order; the gather sums
// a bin front to back, so without this the input gradient differs in its last bit from run to run (advisor, round 5).  Bins hold a handful of entries
// (a tap plane's points with a corner on this input pixel): an insertion sort by the one thread that owns the bin.
__device__ __forceinline__ bool csr_before(const CsrEntry& a, const CsrEntry& b) {
  return a.src < b.src || (a.src == b.src && __float_as_uint(a.w) < __float_as_uint(b.w));
}
__device__ __forceinline__ void csr_sort_bin(CsrEntry* __restrict__ e, int n) {
  if (n <= 8) {                                         // the usual case: the whole bin in registers -- one round trip of loads, a 19-comparator network, one of stores
    unsigned long long k[8];                            // (a chain of dependent global accesses, as in the loop below, costs ~2 us per step)
    const unsigned long long* __restrict__ q = reinterpret_cast<const unsigned long long*>(e);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned long long v = i < n ? q[i] : ~0ull;                       // {src, w} as stored: src in the low word
      k[i] = i < n ? ((v << 32) | (v >> 32)) : ~0ull;                          // key: src, then the weight's bits
    }
#define PRN_CSR_CX(I, J) { const unsigned long long lo = k[I] < k[J] ? k[I] : k[J], hi = k[I] < k[J] ? k[J] : k[I]; k[I] = lo; k[J] = hi; }
    PRN_CSR_CX(0, 2) PRN_CSR_CX(1, 3) PRN_CSR_CX(4, 6) PRN_CSR_CX(5, 7) PRN_CSR_CX(0, 4) PRN_CSR_CX(1, 5) PRN_CSR_CX(2, 6) PRN_CSR_CX(3, 7)
    PRN_CSR_CX(0, 1) PRN_CSR_CX(2, 3) PRN_CSR_CX(4, 5) PRN_CSR_CX(6, 7) PRN_CSR_CX(2, 4) PRN_CSR_CX(3, 5) PRN_CSR_CX(1, 4) PRN_CSR_CX(3, 6)
    PRN_CSR_CX(1, 2) PRN_CSR_CX(3, 4) PRN_CSR_CX(5, 6)
#undef PRN_CSR_CX
    unsigned long long* __restrict__ o = reinterpret_cast<unsigned long long*>(e);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < n) o[i] = (k[i] << 32) | (k[i] >> 32);
    return;
  }
  for (int i = 1; i < n; ++i) {
    const CsrEntry v = e[i];
    int j = i - 1;
    while (j >= 0 && csr_before(v, e[j])) { e[j + 1] = e[j]; --j; }
    e[j + 1] = v;
  }
}
__global__ __launch_bounds__(256) void dcn_csr_sort_kernel(const int* __restrict__ starts, const int* __restrict__ counts, CsrEntry* __restrict__ entries, int nbins) {
  const int bin = blockIdx.x * 256 + threadIdx.x;
  if (bin < nbins && counts[bin] > 1) csr_sort_bin(entries + starts[bin], counts[bin]);
}

__global__ __launch_bounds__(256) void dcn_csr_count_kernel(OmView om, int* __restrict__ counts, int B, int H, int W,
                                                            int Ho, int Wo, int stride) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int plane = Ho * Wo;
  if (gid >= (int64_t)B * 9 * plane) return;
  const int pix = gid % plane, k = (gid / plane) % 9, b = gid / (9 * plane);
  const int ho = pix / Wo, wo = pix - ho * Wo;
  const Tap t = make_tap(om, b, k, ho, wo, Ho, Wo, H, W, stride, false);
  int* cb = counts + (size_t)(b * 9 + k) * H * W;
  if (t.w00 != 0.f) atomicAdd(cb + t.i00, 1);
  if (t.w01 != 0.f) atomicAdd(cb + t.i01, 1);
  if (t.w10 != 0.f) atomicAdd(cb + t.i10, 1);
  if (t.w11 != 0.f) atomicAdd(cb + t.i11, 1);
}

// exclusive scan of the bin counts in two launches: per-block totals (2048 bins per block), then every block adds the
// totals of the blocks before it to its own in-block scan.  The counts are reset to zero for use as fill cursors.
constexpr int SCAN_PER_THREAD = 8, SCAN_PER_BLOCK = 256 * SCAN_PER_THREAD;

__device__ __forceinline__ int block_sum_256(int v, int* sm) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  const int t = sm[0] + sm[1] + sm[2] + sm[3];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(256) void dcn_csr_block_sums_kernel(const int* __restrict__ counts, int* __restrict__ bsum, int n) {
  __shared__ int sm[4];
  const int base = blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * SCAN_PER_THREAD;
  int v = 0;
#pragma unroll
  for (int i = 0; i < SCAN_PER_THREAD; ++i) v += (base + i < n) ? counts[base + i] : 0;
  const int t = block_sum_256(v, sm);
  if (threadIdx.x == 0) bsum[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void dcn_csr_scan_kernel(int* __restrict__ counts, int* __restrict__ starts, const int* __restrict__ bsum,
                                                           int n) {
  __shared__ int sm[4];
  __shared__ int wave_tot[4];
  int pre = 0;                                          // total of all earlier blocks
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) pre += bsum[i];
  pre = block_sum_256(pre, sm);
  const int base = blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * SCAN_PER_THREAD;
  int c[SCAN_PER_THREAD], mine = 0;
#pragma unroll
  for (int i = 0; i < SCAN_PER_THREAD; ++i) { c[i] = (base + i < n) ? counts[base + i] : 0; mine += c[i]; }
  // exclusive scan of `mine` across the block: wave scan + wave totals
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int incl = mine;
  for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
  if (lane == 63) wave_tot[wv] = incl;
  __syncthreads();
  int off = pre + incl - mine;
  for (int w = 0; w < wv; ++w) off += wave_tot[w];
#pragma unroll
  for (int i = 0; i < SCAN_PER_THREAD; ++i)
    if (base + i < n) { starts[base + i] = off; off += c[i]; counts[base + i] = 0; }
}

__global__ __launch_bounds__(256) void dcn_csr_fill_kernel(OmView om, const int* __restrict__ starts, int* __restrict__ cursor,
                                                           CsrEntry* __restrict__ entries, int B, int H, int W, int Ho, int Wo, int stride) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int plane = Ho * Wo;
  if (gid >= (int64_t)B * 9 * plane) return;
  const int pix = gid % plane, k = (gid / plane) % 9, b = gid / (9 * plane);
  const int ho = pix / Wo, wo = pix - ho * Wo;
  const Tap t = make_tap(om, b, k, ho, wo, Ho, Wo, H, W, stride, false);
  const size_t bb = (size_t)(b * 9 + k) * H * W;
  const int src = k * plane + pix;
  auto put = [&](int idx, float w) {
    if (w != 0.f) {
      const int pos = starts[bb + idx] + atomicAdd(cursor + bb + idx, 1);
      entries[pos] = CsrEntry{src, w * t.mod};
    }
  };
  put(t.i00, t.w00); put(t.i01, t.w01); put(t.i10, t.w10); put(t.i11, t.w11);
}

// The same CSR structure in ONE launch (round 5): workgroup (b, k) owns the H*W bins of its tap plane in LDS -- count (LDS atomics), exclusive
// scan, fill (the LDS words turned into cursors) -- and its entries go to a FIXED segment of the entry buffer, [(b*9+k) * 4*plane, +4*plane): a
// plane's points have at most four corners each, so no scan across planes is needed and `starts` is written directly.  Replaces memset + count +
// block sums + scan + fill (five launches of ~5 us, i.e. five launch boundaries on the input-gradient chain per DCN layer).  H*W <= CSR1_MAX_BINS.
constexpr int CSR1_THREADS = 1024;
constexpr int CSR1_MAX_BINS = 15360;                    // 60 KB of LDS words (+ the scan scratch): 30x40 .. 96x160 input maps
static_assert(CSR1_MAX_BINS * 4 + (CSR1_THREADS / 64) * 4 <= 64 * 1024, "the bins and the scan scratch of one tap plane must fit gfx950's 64 KB of LDS per workgroup");

__global__ __launch_bounds__(CSR1_THREADS) void dcn_csr_build_kernel(OmView om, int* __restrict__ starts, int* __restrict__ counts,
                                                                      CsrEntry* __restrict__ entries, int H, int W, int Ho, int Wo, int stride) {
  extern __shared__ int csr_bins[];                     // [HW] counts, then cursors
  __shared__ int wave_tot[CSR1_THREADS / 64];
  const int bk = blockIdx.x, b = bk / 9, k = bk - b * 9;
  const int HW = H * W, plane = Ho * Wo, tid = threadIdx.x;
  for (int i = tid; i < HW; i += CSR1_THREADS) csr_bins[i] = 0;
  __syncthreads();
  for (int p = tid; p < plane; p += CSR1_THREADS) {
    const int ho = p / Wo, wo = p - ho * Wo;
    const Tap t = make_tap(om, b, k, ho, wo, Ho, Wo, H, W, stride, false);
    if (t.w00 != 0.f) atomicAdd(csr_bins + t.i00, 1);
    if (t.w01 != 0.f) atomicAdd(csr_bins + t.i01, 1);
    if (t.w10 != 0.f) atomicAdd(csr_bins + t.i10, 1);
    if (t.w11 != 0.f) atomicAdd(csr_bins + t.i11, 1);
  }
  __syncthreads();
  // exclusive scan: thread t owns bins [t * per, (t + 1) * per)
  const int per = (HW + CSR1_THREADS - 1) / CSR1_THREADS;
  const int b0 = tid * per, b1 = min(HW, b0 + per);
  int mine = 0;
  for (int i = b0; i < b1; ++i) mine += csr_bins[i];
  const int lane = tid & 63, wv = tid >> 6;
  int incl = mine;
  for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
  if (lane == 63) wave_tot[wv] = incl;
  __syncthreads();
  int off = incl - mine;
  for (int w = 0; w < wv; ++w) off += wave_tot[w];
  const int seg = bk * 4 * plane;                       // first entry of this plane's segment
  const size_t gb = (size_t)bk * HW;
  for (int i = b0; i < b1; ++i) {
    const int c = csr_bins[i];
    starts[gb + i] = seg + off;
    counts[gb + i] = c;
    csr_bins[i] = off;                                  // cursor (segment-relative)
    off += c;
  }
  __syncthreads();
  for (int p = tid; p < plane; p += CSR1_THREADS) {
    const int ho = p / Wo, wo = p - ho * Wo;
    const Tap t = make_tap(om, b, k, ho, wo, Ho, Wo, H, W, stride, false);
    const int src = k * plane + p;
    if (t.w00 != 0.f) entries[seg + atomicAdd(csr_bins + t.i00, 1)] = CsrEntry{src, t.w00 * t.mod};
    if (t.w01 != 0.f) entries[seg + atomicAdd(csr_bins + t.i01, 1)] = CsrEntry{src, t.w01 * t.mod};
    if (t.w10 != 0.f) entries[seg + atomicAdd(csr_bins + t.i10, 1)] = CsrEntry{src, t.w10 * t.mod};
    if (t.w11 != 0.f) entries[seg + atomicAdd(csr_bins + t.i11, 1)] = CsrEntry{src, t.w11 * t.mod};
  }
  __syncthreads();                                      // (this workgroup wrote every entry of its segment; the cursors now hold the bins' ends)
  for (int i = b0; i < b1; ++i) {
    const int beg = i > 0 ? csr_bins[i - 1] : 0, n = csr_bins[i] - beg;
    if (n > 1) csr_sort_bin(entries + seg + beg, n);
  }
}

template <int CH>
__global__ __launch_bounds__(256) void dcn_dx_gather_kernel(const float* __restrict__ dcols, const int* __restrict__ starts,
                                                            const int* __restrict__ counts, const CsrEntry* __restrict__ entries,
                                                            float* __restrict__ dx, int C, int HW, int plane) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= HW) return;
  const int b = blockIdx.z, c0 = blockIdx.y * CH;
  const size_t cstride = (size_t)9 * plane;
  const float* __restrict__ dp = dcols + ((size_t)b * C + c0) * cstride;
  float acc[CH];
#pragma unroll
  for (int cc = 0; cc < CH; ++cc) acc[cc] = 0.f;
  // The walk is a dependent chain (bin -> entry -> CH gathers).  The nine bins are read up front and the next entry is
  // fetched while the gathers of the current one are in flight, so a step of the chain costs one memory round trip, not two.
  int st[9], cn[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const size_t bin = (size_t)(b * 9 + k) * HW + q;
    st[k] = starts[bin];
    cn[k] = counts[bin];
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int s = st[k], n = cn[k];
    CsrEntry nxt;
    nxt.src = 0; nxt.w = 0.f;
    if (n > 0) nxt = entries[s];
    for (int e = s; e < s + n; ++e) {
      const CsrEntry en = nxt;
      if (e + 1 < s + n) nxt = entries[e + 1];
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) acc[cc] += en.w * dp[cc * cstride + en.src];
    }
  }
  float* out = dx + ((size_t)b * C + c0) * HW + q;
#pragma unroll
  for (int cc = 0; cc < CH; ++cc)
    if (c0 + cc < C) out[(size_t)cc * HW] = acc[cc];
}

}  // namespace

static int channel_groups(int C, int64_t threads) {
  static const int forced = prn_env_int("PRN_DCN_GROUPS", 0);                               // PRN_DCN_GROUPS forces the group count (tuning)
  int g = forced > 0 ? forced : (int)(1 + (256 * 8 * 256) / (threads > 0 ? threads : 1));     // aim for ~8 blocks of 256 threads per CU
  if (g > 16) g = 16;
  if (g > C) g = C;
  while (g < C && C % g) ++g;                           // next divisor of C upwards (rounding down left the 30x40 stage at 4 groups)
  return g < 1 ? 1 : g;
}

namespace {
OmView raw_view(const float* om, int Ho, int Wo, float max_offset) {
  OmView v;
  v.off = om; v.msk = nullptr; v.off_bs = v.msk_bs = (int64_t)27 * Ho * Wo; v.maxoff = max_offset; v.raw = 1; v.pad = 1;
  return v;
}
OmView desc_view(const prn_dcn_desc* d, const float* offset, const float* mask) {
  if (d->raw) { OmView v = raw_view(offset, d->Ho, d->Wo, d->max_offset); v.pad = d->pad; return v; }
  OmView v;
  v.off = offset; v.msk = mask; v.off_bs = (int64_t)18 * d->Ho * d->Wo; v.msk_bs = (int64_t)9 * d->Ho * d->Wo; v.maxoff = 0.f; v.raw = 0; v.pad = d->pad;
  return v;
}
}  // namespace

extern "C" int prn_dcn_sample(const float* x, const float* om, float* cols, int B, int C, int H, int W, int Ho, int Wo,
                              int stride, float max_offset, void* stream) {
  PRN_REQUIRE(x && om && cols && B > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "prn_dcn_sample: bad arguments");
  PRN_REQUIRE(Ho == (H + 2 - 3) / stride + 1 && Wo == (W + 2 - 3) / stride + 1, "prn_dcn_sample: output size mismatch");
  const int64_t n = (int64_t)B * 9 * Ho * Wo;
  hipLaunchKernelGGL(dcn_sample_kernel, dim3(cdiv(n, 256), channel_groups(C, n)), dim3(256), 0, (hipStream_t)stream, x, raw_view(om, Ho, Wo, max_offset),
                     cols, B, C, H, W, Ho, Wo, stride);
  PRN_CHECK_LAUNCH("prn_dcn_sample");
  return 0;
}

namespace {
struct BwdWs { int64_t part, counts, starts, bsum, entries, total; int nbins, nblocks; };
BwdWs bwd_ws_layout(int B, int C, int H, int W, int Ho, int Wo) {
  auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
  BwdWs l;
  l.nbins = B * 9 * H * W;
  l.nblocks = cdiv(l.nbins, SCAN_PER_BLOCK);
  l.part = 0;
  l.counts = up((int64_t)channel_groups(C, (int64_t)B * 9 * Ho * Wo) * B * 27 * Ho * Wo * 4);
  l.starts = l.counts + up((int64_t)l.nbins * 4);
  l.bsum = l.starts + up((int64_t)l.nbins * 4);
  l.entries = l.bsum + up((int64_t)l.nblocks * 4);
  l.total = l.entries + (int64_t)B * 9 * Ho * Wo * 4 * sizeof(CsrEntry);
  return l;
}

// d_om (or d_offset + d_mask) from the column gradient: fixed-order channel-group partials, then their sum
int launch_dom(const float* x, const OmView& v, const float* dcols, float* d_off, float* d_msk, char* wsb, const BwdWs& l, int B, int C, int H, int W,
               int Ho, int Wo, int stride, hipStream_t st) {
  const int64_t n = (int64_t)B * 9 * Ho * Wo;
  const int G = channel_groups(C, n);
  hipLaunchKernelGGL(dcn_dom_partial_kernel, dim3(cdiv(n, 256), G), dim3(256), 0, st, x, v, dcols, (float*)(wsb + l.part), B, C, H, W, Ho, Wo, stride);
  PRN_CHECK_LAUNCH("dcn d_om partial");
  const int64_t nom = (int64_t)B * 27 * Ho * Wo;
  hipLaunchKernelGGL(dcn_dom_final_kernel, dim3(cdiv(nom, 256)), dim3(256), 0, st, (const float*)(wsb + l.part), d_off, d_msk, nom, G, Ho * Wo);
  PRN_CHECK_LAUNCH("dcn d_om final");
  return 0;
}

// d-input: invert the scatter into CSR bins, then gather
int launch_dx(const OmView& v, const float* dcols, float* dx, char* wsb, const BwdWs& l, int B, int C, int H, int W, int Ho, int Wo, int stride,
              hipStream_t st) {
  const int64_t n = (int64_t)B * 9 * Ho * Wo;
  int* counts = (int*)(wsb + l.counts);
  int* starts = (int*)(wsb + l.starts);
  int* bsum = (int*)(wsb + l.bsum);
  CsrEntry* entries = (CsrEntry*)(wsb + l.entries);
  static const int one = prn_env_int("PRN_DCN_CSR1", 1);                                   // PRN_DCN_CSR1=0: the five-launch construction (A/B)
  if (one && H * W <= CSR1_MAX_BINS) {
    hipLaunchKernelGGL(dcn_csr_build_kernel, dim3(B * 9), dim3(CSR1_THREADS), (size_t)H * W * sizeof(int), st, v, starts, counts, entries, H, W, Ho, Wo, stride);
  } else {
    if (hipMemsetAsync(counts, 0, (size_t)l.nbins * 4, st) != hipSuccess) { prn_set_error("dcn d-input: memset failed"); return 1; }
    hipLaunchKernelGGL(dcn_csr_count_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, v, counts, B, H, W, Ho, Wo, stride);
    hipLaunchKernelGGL(dcn_csr_block_sums_kernel, dim3(l.nblocks), dim3(256), 0, st, (const int*)counts, bsum, l.nbins);
    hipLaunchKernelGGL(dcn_csr_scan_kernel, dim3(l.nblocks), dim3(256), 0, st, counts, starts, (const int*)bsum, l.nbins);
    hipLaunchKernelGGL(dcn_csr_fill_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, v, (const int*)starts, counts, entries, B, H, W, Ho, Wo, stride);
    hipLaunchKernelGGL(dcn_csr_sort_kernel, dim3(cdiv(l.nbins, 256)), dim3(256), 0, st, (const int*)starts, (const int*)counts, entries, (int)l.nbins);
  }
  PRN_CHECK_LAUNCH("dcn d-input csr");
  const int HW = H * W;
  // channels per thread: 8, or 4 when that leaves fewer than ~4 blocks per CU
  const int64_t blocks8 = (int64_t)cdiv(HW, 256) * cdiv(C, 8) * B;
  if (blocks8 >= 1024)
    hipLaunchKernelGGL((dcn_dx_gather_kernel<8>), dim3(cdiv(HW, 256), cdiv(C, 8), B), dim3(256), 0, st, dcols, (const int*)starts, (const int*)counts,
                       (const CsrEntry*)entries, dx, C, HW, Ho * Wo);
  else
    hipLaunchKernelGGL((dcn_dx_gather_kernel<4>), dim3(cdiv(HW, 256), cdiv(C, 4), B), dim3(256), 0, st, dcols, (const int*)starts, (const int*)counts,
                       (const CsrEntry*)entries, dx, C, HW, Ho * Wo);
  PRN_CHECK_LAUNCH("dcn d-input gather");
  return 0;
}
}  // namespace

extern "C" int64_t prn_dcn_sample_bwd_ws_bytes(int B, int C, int H, int W, int Ho, int Wo) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || (int64_t)B * 9 * H * W >= (1LL << 31)) return -1;
  return bwd_ws_layout(B, C, H, W, Ho, Wo).total;
}

extern "C" int prn_dcn_sample_bwd(const float* x, const float* om, const float* dcols, float* dx, float* d_om, void* ws,
                                  int B, int C, int H, int W, int Ho, int Wo, int stride, float max_offset, void* stream) {
  PRN_REQUIRE(x && om && dcols && dx && d_om && ws && B > 0 && C > 0, "prn_dcn_sample_bwd: bad arguments");
  PRN_REQUIRE((int64_t)B * 9 * H * W < (1LL << 31) && (int64_t)B * 36 * Ho * Wo < (1LL << 31), "prn_dcn_sample_bwd: map too large");
  hipStream_t st = (hipStream_t)stream;
  const BwdWs l = bwd_ws_layout(B, C, H, W, Ho, Wo);
  const OmView v = raw_view(om, Ho, Wo, max_offset);
  if (int e = launch_dom(x, v, dcols, d_om, nullptr, (char*)ws, l, B, C, H, W, Ho, Wo, stride, st)) return e;
  return launch_dx(v, dcols, dx, (char*)ws, l, B, C, H, W, Ho, Wo, stride, st);
}

// ---- the data-gradient half of the fused operator (include/prn.h: prn_dcnv2_bwd_input / prn_dcnv2_bwd_offset_mask) ----------
// Workspace: [ dcols = W^T dy : B * 9C * Ho * Wo floats ][ split-K partials of that GEMM ][ d_om partials + CSR (bwd_ws_layout) ]
namespace {
struct DataWs { int64_t dcols, gemm, rest, total; prn_conv_desc g; };
int data_ws_layout(const prn_dcn_desc* d, DataWs& l) {
  auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
  prn_conv_desc& g = l.g;                                  // dcols[9C x N] = wt[9C x M] * dy[M x N]: a 1x1 convolution of dy
  g.B = d->B; g.C = d->M; g.H = d->Ho; g.W = d->Wo; g.M = d->C * 9; g.KH = g.KW = 1; g.stride = 1; g.pad = 0; g.Ho = d->Ho; g.Wo = d->Wo;
  g.in_mode = PRN_IN_ZERO; g.dil = 1; g.epilogue = PRN_EPI_NONE; g.ystride = 0; g.yH = g.yW = 0; g.reserved = 0;
  g.opts = d->opts;
  const int64_t gb = prn_conv2d_fwd_ws_bytes(&g);
  if (gb < 0) return 2;
  if ((int64_t)d->B * 9 * d->H * d->W >= (1LL << 31) || (int64_t)d->B * 36 * d->Ho * d->Wo >= (1LL << 31)) { prn_set_error("prn_dcnv2_bwd: map too large"); return 2; }
  l.dcols = 0;
  l.gemm = up((int64_t)d->B * d->C * 9 * d->Ho * d->Wo * 4);
  l.rest = l.gemm + up(gb);
  l.total = l.rest + bwd_ws_layout(d->B, d->C, d->H, d->W, d->Ho, d->Wo).total;
  return 0;
}
int check_dcn_desc(const prn_dcn_desc* d, const char* who) {
  PRN_REQUIRE(d != nullptr && d->B > 0 && d->C > 0 && d->H > 0 && d->W > 0 && d->M > 0 && d->Ho > 0 && d->Wo > 0 && d->stride > 0 && d->pad >= 0,
              "%s: bad descriptor", who);
  PRN_REQUIRE(d->Ho == (d->H + 2 * d->pad - 3) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - 3) / d->stride + 1, "%s: output size mismatch", who);
  return 0;
}
}  // namespace

extern "C" int64_t prn_dcnv2_bwd_ws_bytes(const prn_dcn_desc* d) {
  if (check_dcn_desc(d, "prn_dcnv2_bwd_ws_bytes")) return -1;
  DataWs l;
  if (data_ws_layout(d, l)) return -1;
  return l.total;
}

extern "C" int prn_dcnv2_bwd_input(const prn_dcn_desc* d, const float* dy, const float* wt, const void* wt_images, const float* offset, const float* mask,
                                   float* dx, void* ws, void* stream) {
  return prn_dcnv2_bwd_input_phase(d, dy, wt, wt_images, offset, mask, dx, ws, stream, 0);
}

// phase 0: everything; 1 / 2: the column-gradient GEMM W^T dy (launch / its K-split sum); 3: the CSR gather of dx from the column
// gradient already in ws (profiler brackets: each launch timed on its own, nothing issued twice)
extern "C" int prn_dcnv2_bwd_input_phase(const prn_dcn_desc* d, const float* dy, const float* wt, const void* wt_images, const float* offset, const float* mask,
                                         float* dx, void* ws, void* stream, int phase) {
  if (int e = check_dcn_desc(d, "prn_dcnv2_bwd_input")) return e;
  PRN_REQUIRE(dy && wt && offset && ws, "prn_dcnv2_bwd_input: null tensor");
  PRN_REQUIRE(phase >= 0 && phase <= 3, "prn_dcnv2_bwd_input: bad phase");
  DataWs l;
  if (int e = data_ws_layout(d, l)) return e;
  char* wsb = (char*)ws;
  if (phase <= 2)
    if (int e = prn_conv2d_fwd_counted(&l.g, dy, wt, wt_images, nullptr, nullptr, (float*)(wsb + l.dcols), wsb + l.gemm, nullptr, stream, phase)) return e;
  if (dx == nullptr || phase == 1 || phase == 2) return 0;   // (dx == NULL: column gradient only, the caller just wants prn_dcnv2_bwd_offset_mask)
  const BwdWs bl = bwd_ws_layout(d->B, d->C, d->H, d->W, d->Ho, d->Wo);
  return launch_dx(desc_view(d, offset, mask), (const float*)(wsb + l.dcols), dx, wsb + l.rest, bl, d->B, d->C, d->H, d->W, d->Ho, d->Wo, d->stride,
                   (hipStream_t)stream);
}

extern "C" int prn_dcnv2_bwd_offset_mask(const prn_dcn_desc* d, const float* x, const float* offset, const float* mask, float* d_offset, float* d_mask,
                                         void* ws, void* stream) {
  if (int e = check_dcn_desc(d, "prn_dcnv2_bwd_offset_mask")) return e;
  PRN_REQUIRE(x && offset && d_offset && ws, "prn_dcnv2_bwd_offset_mask: null tensor");
  PRN_REQUIRE(d->raw || d_mask != nullptr, "prn_dcnv2_bwd_offset_mask: d_mask required unless the descriptor is raw (then d_offset is the [B,27,Ho,Wo] gradient)");
  DataWs l;
  if (int e = data_ws_layout(d, l)) return e;
  char* wsb = (char*)ws;
  const BwdWs bl = bwd_ws_layout(d->B, d->C, d->H, d->W, d->Ho, d->Wo);
  return launch_dom(x, desc_view(d, offset, mask), (const float*)(wsb + l.dcols), d_offset, d->raw ? nullptr : d_mask, wsb + l.rest, bl, d->B, d->C, d->H,
                    d->W, d->Ho, d->Wo, d->stride, (hipStream_t)stream);
}

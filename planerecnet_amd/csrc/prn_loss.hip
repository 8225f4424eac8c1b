// Fused reductions of the joint loss (reference models/functions/losses.py).  The reference -- and round 1 of this build --
// evaluate each term as a chain of elementwise / reduction operators over the same [instances x 120 x 160] tensor (sigmoid,
// three products, three row sums, divisions, index_add, ... ~15 passes forward, as many backward); here a term is one pass
// forward and one pass backward.
#include "prn_common.h"

namespace {

// ---- instance masks: Dice (losses.py:69-118,355-368) + lava (losses.py:169-197,288-329) ----------------------------------
// rows i < P: one positive grid cell = one predicted mask (logits z[i][HW], sigmoid applied here), target t[i][HW] (uint8 0/1),
// image img[i]; adj[b][HW] = the image's depth-gradient map pulled back to mask resolution (GT only).
//   a = sum p t, b = sum p^2, c = sum t^2, l = sum p adj          per row, in fixed-order partials
//   dice_i = 1 - 2a / (b + c + 0.002)                              ins = w_ins * mean_i dice_i
//   lav    = w_lav * mean over qualifying images of  sum_{i in image} l_i / (gsum_b * npos_b)
constexpr int ML_SPLITS = 4;

__global__ __launch_bounds__(256) void mask_loss_partial_kernel(const float* __restrict__ z, const unsigned char* __restrict__ t,
                                                                const float* __restrict__ adj, const int64_t* __restrict__ img,
                                                                float* __restrict__ part, int HW4) {
  const int i = blockIdx.x, s = blockIdx.y;
  const float4* zr = reinterpret_cast<const float4*>(z) + (size_t)i * HW4;
  const uchar4* tr = reinterpret_cast<const uchar4*>(t) + (size_t)i * HW4;
  const float4* ar = adj ? reinterpret_cast<const float4*>(adj) + (size_t)img[i] * HW4 : nullptr;
  const int beg = (int)((int64_t)HW4 * s / ML_SPLITS), end = (int)((int64_t)HW4 * (s + 1) / ML_SPLITS);
  float a = 0.f, b = 0.f, c = 0.f, l = 0.f;
  for (int q = beg + threadIdx.x; q < end; q += 256) {
    const float4 v = zr[q];
    const uchar4 u = tr[q];
    const float p0 = 1.f / (1.f + expf(-v.x)), p1 = 1.f / (1.f + expf(-v.y)), p2 = 1.f / (1.f + expf(-v.z)), p3 = 1.f / (1.f + expf(-v.w));
    const float t0 = u.x, t1 = u.y, t2 = u.z, t3 = u.w;
    a += (p0 * t0 + p1 * t1) + (p2 * t2 + p3 * t3);
    b += (p0 * p0 + p1 * p1) + (p2 * p2 + p3 * p3);
    c += (t0 * t0 + t1 * t1) + (t2 * t2 + t3 * t3);
    if (ar) { const float4 g = ar[q]; l += (p0 * g.x + p1 * g.y) + (p2 * g.z + p3 * g.w); }
  }
  __shared__ float sm[4][4];
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); l = wave_sum(l);
  if ((threadIdx.x & 63) == 0) { float* r = sm[threadIdx.x >> 6]; r[0] = a; r[1] = b; r[2] = c; r[3] = l; }
  __syncthreads();
  if (threadIdx.x < 4) part[((size_t)i * ML_SPLITS + s) * 4 + threadIdx.x] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// one workgroup: rows -> the two scalars + the per-row coefficients of the backward pass
//   coef[i] = { dA, dB, dL }:  d(ins)/dp = dA * t + dB * p,   d(lav)/dp = dL * adj
__global__ __launch_bounds__(256) void mask_loss_final_kernel(const float* __restrict__ part, const int64_t* __restrict__ img,
                                                              const float* __restrict__ gsum, const float* __restrict__ npos, float* __restrict__ out,
                                                              float* __restrict__ coef, int P, int B, float w_ins, float w_lav) {
  __shared__ double num[64];          // per-image lava numerators (B <= 64)
  __shared__ double red[256];
  __shared__ int nok;
  for (int b = threadIdx.x; b < 64; b += 256) num[b] = 0.0;
  __syncthreads();
  double dice = 0.0;
  for (int i = threadIdx.x; i < P; i += 256) {
    double a = 0, bb = 0, c = 0;
    for (int s = 0; s < ML_SPLITS; ++s) { const float* r = part + ((size_t)i * ML_SPLITS + s) * 4; a += r[0]; bb += r[1]; c += r[2]; }
    const double D = bb + 0.001 + c + 0.001;
    dice += 1.0 - 2.0 * a / D;
    coef[(size_t)i * 3 + 0] = (float)(-2.0 / D * w_ins / P);
    coef[(size_t)i * 3 + 1] = (float)(4.0 * a / (D * D) * w_ins / P);
  }
  red[threadIdx.x] = dice;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (gsum) {
    // lava numerators: one wave per image, lanes over the rows (any row order), a shuffle tree -- fixed order; the one-thread loop over P * ML_SPLITS
    // dependent loads that stood here took 75 us of the loss phase
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b = wave; b < B && b < 64; b += 4) {
      double acc = 0.0;
      for (int i = lane; i < P; i += 64) {
        if ((int)img[i] != b) continue;
        double l = 0;
        for (int s = 0; s < ML_SPLITS; ++s) l += part[((size_t)i * ML_SPLITS + s) * 4 + 3];
        acc += l;
      }
      acc = wave_sum_d(acc);
      if (lane == 0) num[b] = acc;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = (float)(red[0] / P * w_ins);
    // lava: rows are sorted by image (the loss lays them out image by image): fixed-order sums
    int ok = 0;
    if (gsum) {
      double tot = 0.0;
      for (int b = 0; b < B; ++b)
        if (gsum[b] > 0.f && npos[b] > 0.f) { ++ok; tot += num[b] / fmax((double)gsum[b] * npos[b], 1e-30); }
      out[1] = (float)(tot / (ok > 0 ? ok : 1) * w_lav);
    } else {
      out[1] = 0.f;
    }
    nok = ok > 0 ? ok : 1;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < P; i += 256) {
    float dl = 0.f;
    if (gsum) {
      const int b = (int)img[i];
      if (gsum[b] > 0.f && npos[b] > 0.f) dl = (float)(w_lav / nok / fmax((double)gsum[b] * npos[b], 1e-30));
    }
    coef[(size_t)i * 3 + 2] = dl;
  }
}

// dz = p (1 - p) * ( g_ins * (dA t + dB p) + g_lav * dL adj )
__global__ __launch_bounds__(256) void mask_loss_bwd_kernel(const float* __restrict__ z, const unsigned char* __restrict__ t,
                                                            const float* __restrict__ adj, const int64_t* __restrict__ img,
                                                            const float* __restrict__ coef, const float* __restrict__ g_ins, const float* __restrict__ g_lav, float* __restrict__ dz,
                                                            int HW4) {
  const int i = blockIdx.x;
  const float4* zr = reinterpret_cast<const float4*>(z) + (size_t)i * HW4;
  const uchar4* tr = reinterpret_cast<const uchar4*>(t) + (size_t)i * HW4;
  const float4* ar = adj ? reinterpret_cast<const float4*>(adj) + (size_t)img[i] * HW4 : nullptr;
  float4* dr = reinterpret_cast<float4*>(dz) + (size_t)i * HW4;
  const float gi = g_ins ? g_ins[0] : 0.f, gl = g_lav ? g_lav[0] : 0.f;
  const float dA = gi * coef[(size_t)i * 3], dB = gi * coef[(size_t)i * 3 + 1], dL = gl * coef[(size_t)i * 3 + 2];
  for (int q = blockIdx.y * 256 + threadIdx.x; q < HW4; q += gridDim.y * 256) {
    const float4 v = zr[q];
    const uchar4 u = tr[q];
    float4 gg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ar) gg = ar[q];
    const float p0 = 1.f / (1.f + expf(-v.x)), p1 = 1.f / (1.f + expf(-v.y)), p2 = 1.f / (1.f + expf(-v.z)), p3 = 1.f / (1.f + expf(-v.w));
    float4 o;
    o.x = p0 * (1.f - p0) * (dA * u.x + dB * p0 + dL * gg.x);
    o.y = p1 * (1.f - p1) * (dA * u.y + dB * p1 + dL * gg.y);
    o.z = p2 * (1.f - p2) * (dA * u.z + dB * p2 + dL * gg.z);
    o.w = p3 * (1.f - p3) * (dA * u.w + dB * p3 + dL * gg.w);
    dr[q] = o;
  }
}

}  // namespace

extern "C" int prn_mask_loss_ws_floats(int P) { return P * ML_SPLITS * 4; }

extern "C" int prn_mask_loss_fwd(const float* logits, const unsigned char* labels, const float* adj, const int64_t* img, const float* gsum,
                                 const float* npos, float* out2, float* coef, float* ws, int P, int HW, int B, float w_ins, float w_lav,
                                 void* stream) {
  PRN_REQUIRE(logits && labels && img && out2 && coef && ws && P > 0 && HW > 0 && (HW & 3) == 0 && B > 0 && B <= 64,
              "prn_mask_loss_fwd: bad arguments (P=%d HW=%d B=%d; HW %% 4 == 0, B <= 64)", P, HW, B);
  PRN_REQUIRE((adj == nullptr) == (gsum == nullptr) && (adj == nullptr || npos != nullptr), "prn_mask_loss_fwd: adj / gsum / npos go together");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mask_loss_partial_kernel, dim3(P, ML_SPLITS), dim3(256), 0, st, logits, labels, adj, img, ws, HW / 4);
  PRN_CHECK_LAUNCH("prn_mask_loss_fwd/partial");
  hipLaunchKernelGGL(mask_loss_final_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, img, gsum, npos, out2, coef, P, B, w_ins, w_lav);
  PRN_CHECK_LAUNCH("prn_mask_loss_fwd/final");
  return 0;
}

extern "C" int prn_mask_loss_bwd(const float* logits, const unsigned char* labels, const float* adj, const int64_t* img, const float* coef,
                                 const float* g_ins, const float* g_lav, float* dlogits, int P, int HW, void* stream) {
  PRN_REQUIRE(logits && labels && img && coef && dlogits && P > 0 && HW > 0 && (HW & 3) == 0, "prn_mask_loss_bwd: bad arguments");
  const int gy = cdiv(HW / 4, 256 * 4) > 0 ? cdiv(HW / 4, 256 * 4) : 1;
  hipLaunchKernelGGL(mask_loss_bwd_kernel, dim3(P, gy), dim3(256), 0, (hipStream_t)stream, logits, labels, adj, img, coef, g_ins, g_lav, dlogits, HW / 4);
  PRN_CHECK_LAUNCH("prn_mask_loss_bwd");
  return 0;
}

// ---- virtual-normal loss (reference models/functions/vnl.py:57-165), per-triplet part --------------------------------------
// One thread per sampled triplet: the three cloud points (pred and GT), the validity filter (vnl.py:71-104), the normals, the
// cosine term -- about 80 elementwise / gather / reduction operators over [n_triplets x 3 x 3] tensors in the
// operator-by-operator form -- and, in the same pass, the three partial derivatives d loss / d depth(point p) by forward-mode
// differentiation (dual numbers with three tangent slots: a triplet's loss depends on exactly three predicted depths), so
// the backward pass is one weighted scatter of those numbers instead of a replay of the whole chain.
namespace {

struct D3 {                        // value + tangents w.r.t. the predicted depth of point 0 / 1 / 2
  float v, d[3];
};
__device__ __forceinline__ D3 dconst(float v) { return D3{v, {0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D3 operator+(const D3& a, const D3& b) { return D3{a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}}; }
__device__ __forceinline__ D3 operator-(const D3& a, const D3& b) { return D3{a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}}; }
__device__ __forceinline__ D3 operator*(const D3& a, const D3& b) {
  return D3{a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2]}};
}
__device__ __forceinline__ D3 operator*(const D3& a, float s) { return D3{a.v * s, {a.d[0] * s, a.d[1] * s, a.d[2] * s}}; }
__device__ __forceinline__ D3 operator/(const D3& a, const D3& b) {          // (a / b)' = (a' - (a / b) b') / b
  const float q = a.v / b.v;
  return D3{q, {(a.d[0] - q * b.d[0]) / b.v, (a.d[1] - q * b.d[1]) / b.v, (a.d[2] - q * b.d[2]) / b.v}};
}
__device__ __forceinline__ D3 dabs(const D3& a) {                             // autograd: sgn(0) = 0
  const float s = a.v > 0.f ? 1.f : (a.v < 0.f ? -1.f : 0.f);
  return D3{fabsf(a.v), {s * a.d[0], s * a.d[1], s * a.d[2]}};
}
// 2-norm of a 3-vector with autograd's subgradient at zero (0)
__device__ __forceinline__ D3 dnorm3(const D3& x, const D3& y, const D3& z) {
  const float n = sqrtf(x.v * x.v + y.v * y.v + z.v * z.v);
  D3 r{n, {0.f, 0.f, 0.f}};
  if (n > 0.f) {
#pragma unroll
    for (int k = 0; k < 3; ++k) r.d[k] = (x.v * x.d[k] + y.v * y.d[k] + z.v * z.d[k]) / n;
  }
  return r;
}
// the same machinery in double for the plane branch (the reference's plane normals are float64: quirk Q6)
struct E3 {
  double v, d[3];
};
__device__ __forceinline__ E3 widen(const D3& a) { return E3{(double)a.v, {(double)a.d[0], (double)a.d[1], (double)a.d[2]}}; }

struct VnlArgs {
  const float* pred; const float* gt;           // [B * H * W] depths
  const int* gid;                                // [3][n] cloud-point index of the triplet's points
  const int64_t* seg;                            // [n] segment (plane / non-planar region of one image) of the triplet
  const unsigned char* seg_is_plane;             // [nseg] (bool)
  const double* seg_normal;                      // [nseg][3]
  const int64_t* seg_img;                        // [nseg]
  const double* fx; const double* fy;            // [B]
  double* loss; unsigned char* valid; float* g3; // [n], [n], [n][3]
  int n, H, W;
  float delta_z;
};

struct P3 { float x, y, z; };

__device__ __forceinline__ bool vnl_filter(const P3 p[3], float delta_diff, float delta_z) {     // vnl.py:71-104
  const float delta_cos = 0.985f;
  float d[3][3];                                // [xyz][pair]: p1 - p0, p2 - p0, p2 - p1
  d[0][0] = p[1].x - p[0].x; d[0][1] = p[2].x - p[0].x; d[0][2] = p[2].x - p[1].x;
  d[1][0] = p[1].y - p[0].y; d[1][1] = p[2].y - p[0].y; d[1][2] = p[2].y - p[1].y;
  d[2][0] = p[1].z - p[0].z; d[2][1] = p[2].z - p[0].z; d[2][2] = p[2].z - p[1].z;
  float qn[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) qn[j] = sqrtf(d[0][j] * d[0][j] + d[1][j] * d[1][j] + d[2][j] * d[2][j]);
  int ncos = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float e = (d[0][i] * d[0][j] + d[1][i] * d[1][j] + d[2][i] * d[2][j]) / (qn[i] * qn[j] + 1e-8f);
      ncos += (e > delta_cos || e < -delta_cos) ? 1 : 0;
    }
  const bool m_cos = ncos > 3;
  const bool m_pad = p[0].z > delta_z && p[1].z > delta_z && p[2].z > delta_z;
  bool near = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) near = near && (fabsf(d[a][0]) < delta_diff || fabsf(d[a][1]) < delta_diff || fabsf(d[a][2]) < delta_diff);
  return m_pad && !(near || m_cos);
}

__global__ __launch_bounds__(256) void vnl_triplet_kernel(VnlArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int HW = a.H * a.W;
  const int64_t sg = a.seg[i];
  const bool plane = a.seg_is_plane[sg] != 0;
  const int64_t b = a.seg_img[sg];
  const float fx = (float)a.fx[b], fy = (float)a.fy[b];
  const float u0 = (float)(a.W / 2), v0 = (float)(a.H / 2);
  P3 pp[3], pg[3];
  D3 X[3], Y[3], Z[3];                          // predicted points as duals
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const int g = a.gid[(size_t)p * a.n + i];
    const int pix = g % HW, v = pix / a.W, u = pix - v * a.W;
    const float uu = (float)u - u0, vv = (float)v - v0;
    const float dp = a.pred[g], dg = a.gt[g];
    D3 d{dp, {0.f, 0.f, 0.f}};
    d.d[p] = 1.f;
    const D3 ad = dabs(d);
    // u * |d| / fx, evaluated as (u * |d|) / fx like the reference (vnl.py:34-41)
    X[p] = D3{uu * ad.v / fx, {uu * ad.d[0] / fx, uu * ad.d[1] / fx, uu * ad.d[2] / fx}};
    Y[p] = D3{vv * ad.v / fy, {vv * ad.d[0] / fy, vv * ad.d[1] / fy, vv * ad.d[2] / fy}};
    Z[p] = d;
    pp[p] = P3{X[p].v, Y[p].v, dp};
    const float ag = fabsf(dg);
    pg[p] = P3{uu * ag / fx, vv * ag / fy, dg};
  }
  // planes filter on the predicted cloud (delta_diff 0.005), the non-planar region on the GT cloud (0.1): vnl.py:131,146
  const bool ok = vnl_filter(plane ? pp : pg, plane ? 0.005f : 0.1f, a.delta_z);
  // non-planar branch, vnl.py:151: `pw_pred[pw_pred[:, 2, :] == 0] = 0.0001` -- a [n,3] mask indexing dims (0,1) of the
  // [n, xyz, point] tensor: point j having z == 0 overwrites COORDINATE j of all three points
  if (!plane) {
    const bool z0 = pp[0].z == 0.f, z1 = pp[1].z == 0.f, z2 = pp[2].z == 0.f;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      if (z0) X[p] = dconst(1e-4f);
      if (z1) Y[p] = dconst(1e-4f);
      if (z2) Z[p] = dconst(1e-4f);
    }
  }
  // predicted normal: cross(p1 - p0, p2 - p0) / (|.| + 0.01 [|.| == 0])
  const D3 e1x = X[1] - X[0], e1y = Y[1] - Y[0], e1z = Z[1] - Z[0], e2x = X[2] - X[0], e2y = Y[2] - Y[0], e2z = Z[2] - Z[0];
  D3 nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
  D3 nn = dnorm3(nx, ny, nz);
  if (nn.v == 0.f) nn.v = 0.01f;                             // (+ 0.01 where the norm is exactly 0; its tangent is 0 there)
  nx = nx / nn; ny = ny / nn; nz = nz / nn;
  double lossv;
  float g[3];
  const double eps = 1e-8;
  if (plane) {
    // |cosine_similarity(dn.double(), plane normal (float64))|: x / max(|x|, eps) . y / max(|y|, eps)
    const E3 x = widen(nx), y = widen(ny), z = widen(nz);
    const double* nr = a.seg_normal + (size_t)sg * 3;
    const double xn = sqrt(x.v * x.v + y.v * y.v + z.v * z.v);
    const double rn = sqrt(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]);
    const double xc = fmax(xn, eps), rc = fmax(rn, eps);
    const double dot = x.v * nr[0] + y.v * nr[1] + z.v * nr[2];
    const double c = dot / (xc * rc);
    const double s = c > 0.0 ? 1.0 : (c < 0.0 ? -1.0 : 0.0);
    lossv = 1.0 - fabs(c);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double ddot = x.d[k] * nr[0] + y.d[k] * nr[1] + z.d[k] * nr[2];
      const double dxn = (xn > eps) ? (x.v * x.d[k] + y.v * y.d[k] + z.v * z.d[k]) / xn : 0.0;   // clamp_min passes the gradient above eps only
      const double dc = (ddot - c * rc * dxn) / (xc * rc);
      g[k] = (float)(-s * dc);
    }
  } else {
    // GT normal (no gradient), cosine similarity in float, then widened
    const float g1x = pg[1].x - pg[0].x, g1y = pg[1].y - pg[0].y, g1z = pg[1].z - pg[0].z;
    const float g2x = pg[2].x - pg[0].x, g2y = pg[2].y - pg[0].y, g2z = pg[2].z - pg[0].z;
    float mx = g1y * g2z - g1z * g2y, my = g1z * g2x - g1x * g2z, mz = g1x * g2y - g1y * g2x;
    float mn = sqrtf(mx * mx + my * my + mz * mz);
    if (mn == 0.f) mn = 0.01f;
    mx /= mn; my /= mn; mz /= mn;
    const float epsf = 1e-8f;
    const float xn = sqrtf(nx.v * nx.v + ny.v * ny.v + nz.v * nz.v), rn = sqrtf(mx * mx + my * my + mz * mz);
    const float xc = fmaxf(xn, epsf), rc = fmaxf(rn, epsf);
    const float dot = nx.v * mx + ny.v * my + nz.v * mz;
    const float c = dot / (xc * rc);
    const float s = c > 0.f ? 1.f : (c < 0.f ? -1.f : 0.f);
    lossv = 1.0 - (double)fabsf(c);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float ddot = nx.d[k] * mx + ny.d[k] * my + nz.d[k] * mz;
      const float dxn = (xn > epsf) ? (nx.v * nx.d[k] + ny.v * ny.d[k] + nz.v * nz.d[k]) / xn : 0.f;
      const float dc = (ddot - c * rc * dxn) / (xc * rc);
      g[k] = -s * dc;
    }
  }
  const bool finite = lossv == lossv;                        // NaN losses are dropped downstream (nansum): no gradient either
  a.loss[i] = lossv;
  a.valid[i] = ok ? 1 : 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) a.g3[(size_t)i * 3 + k] = (finite && g[k] == g[k]) ? g[k] : 0.f;
}

// d depth[r] = sum over the (triplet, point) pairs that sampled cloud point r of  g_loss[triplet] * g3[triplet][point]
// `order` = argsort of the flattened [3][n] index array (GT only, once per step), `start` = exclusive prefix sum of the per-point
// counts: fixed order, no atomics.
__global__ __launch_bounds__(256) void vnl_scatter_kernel(const double* __restrict__ gl, const float* __restrict__ g3, const int64_t* __restrict__ order,
                                                          const int64_t* __restrict__ start, float* __restrict__ dd, int npts, int n) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= npts) return;
  const int64_t s0 = start[r], s1 = start[r + 1];
  double acc = 0.0;
  for (int64_t k = s0; k < s1; ++k) {
    const int64_t e = order[k];
    const int64_t p = e / n, t = e - p * n;
    acc += gl[t] * (double)g3[t * 3 + p];
  }
  dd[r] = (float)acc;
}

}  // namespace

extern "C" int prn_vnl_triplets(const float* pred, const float* gt, const int* gid, const int64_t* seg, const unsigned char* seg_is_plane,
                                const double* seg_normal, const int64_t* seg_img, const double* fx, const double* fy, double* loss,
                                unsigned char* valid, float* g3, int n, int H, int W, float delta_z, void* stream) {
  PRN_REQUIRE(pred && gt && gid && seg && seg_is_plane && seg_normal && seg_img && fx && fy && loss && valid && g3 && n > 0 && H > 0 && W > 0,
              "prn_vnl_triplets: bad arguments");
  VnlArgs a{pred, gt, gid, seg, seg_is_plane, seg_normal, seg_img, fx, fy, loss, valid, g3, n, H, W, delta_z};
  hipLaunchKernelGGL(vnl_triplet_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a);
  PRN_CHECK_LAUNCH("prn_vnl_triplets");
  return 0;
}

extern "C" int prn_vnl_scatter(const double* g_loss, const float* g3, const int64_t* order, const int64_t* start, float* d_depth, int npts, int n,
                               void* stream) {
  PRN_REQUIRE(g_loss && g3 && order && start && d_depth && npts > 0 && n > 0, "prn_vnl_scatter: bad arguments");
  hipLaunchKernelGGL(vnl_scatter_kernel, dim3(cdiv(npts, 256)), dim3(256), 0, (hipStream_t)stream, g_loss, g3, order, start, d_depth, npts, n);
  PRN_CHECK_LAUNCH("prn_vnl_scatter");
  return 0;
}

// ---------------------------------------------------------------------------------------------- focal and RMSE-log terms
// Category term (models/functions/losses.py:121-138,331-352): sigmoid focal loss summed over all cells and classes, and the depth
// term (losses.py:142-147,371-392): per image sqrt(mean over valid pixels of (log pred - log gt)^2), averaged over the images.  Each
// was ~15 / ~10 elementwise + reduction launches forward and as many backward; here one pass each way, fp64 partials in a fixed order.
namespace {
constexpr int RED_BLOCKS = 128;

__device__ __forceinline__ void focal_terms(float x, bool pos, float alpha, float gamma, float& loss, float& dldx) {
  // ce = -log p_t (binary_cross_entropy_with_logits' stable form), p_t = p if pos else 1 - p
  const float ax = fabsf(x), l1p = log1pf(expf(-ax));
  const float ce = fmaxf(x, 0.f) - (pos ? x : 0.f) + l1p;
  const float p = 1.f / (1.f + expf(-x));
  const float pt = pos ? p : 1.f - p, q = 1.f - pt;
  const float w = alpha >= 0.f ? (pos ? alpha : 1.f - alpha) : 1.f;
  const float qg = gamma == 2.f ? q * q : powf(q, gamma), qg1 = gamma == 2.f ? q : powf(q, gamma - 1.f);
  loss = w * ce * qg;
  // d/dp_t [ -log(p_t) q^g ] = -q^g / p_t - g q^(g-1) (-log p_t) ... with -log p_t = ce ; d p_t / dx = +-p(1-p)
  const float dpt = -qg / fmaxf(pt, 1e-38f) + gamma * qg1 * (-ce);
  dldx = w * dpt * (pos ? 1.f : -1.f) * p * (1.f - p);
}

__global__ __launch_bounds__(256) void focal_partial_kernel(const float* __restrict__ x, const int64_t* __restrict__ label, double* __restrict__ part,
                                                            int64_t n, int C, float alpha, float gamma) {
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float l, d;
    focal_terms(x[i], label[i / C] == (int64_t)(i % C), alpha, gamma, l, d);
    acc += (double)l;
  }
  __shared__ double sm[4];
  const double v = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ void sum_partials_kernel(const double* __restrict__ part, int nb, float* __restrict__ out) {      // one wave: lanes over the partial sums, a shuffle tree
  double t = 0.0;
  for (int b = threadIdx.x; b < nb; b += 64) t += part[b];
  t = wave_sum_d(t);
  if (threadIdx.x == 0) out[0] = (float)t;
}

__global__ __launch_bounds__(256) void focal_bwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ label, const float* __restrict__ g,
                                                        float* __restrict__ dx, int64_t n, int C, float alpha, float gamma) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float l, d;
  focal_terms(x[i], label[i / C] == (int64_t)(i % C), alpha, gamma, l, d);
  dx[i] = g[0] * d;
}

// per image b and block: sum over valid pixels of (log max(pred, clamp) - log max(gt, clamp))^2, and the number of valid pixels
__global__ __launch_bounds__(256) void rmse_log_partial_kernel(const float* __restrict__ pred, const float* __restrict__ gt, double* __restrict__ part,
                                                               int HW, float min_depth, float clamp) {
  const int b = blockIdx.y;
  const float* p = pred + (size_t)b * HW;
  const float* g = gt + (size_t)b * HW;
  double s = 0.0, c = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const float gv = g[i];
    if (gv > min_depth) {
      const float d = logf(fmaxf(p[i], clamp)) - logf(fmaxf(gv, clamp));
      s += (double)(d * d);
      c += 1.0;
    }
  }
  __shared__ double sm[4][2];
  const double vs = wave_sum_d(s), vc = wave_sum_d(c);
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6][0] = vs; sm[threadIdx.x >> 6][1] = vc; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[((size_t)b * gridDim.x + blockIdx.x) * 2] = (sm[0][0] + sm[1][0]) + (sm[2][0] + sm[3][0]);
    part[((size_t)b * gridDim.x + blockIdx.x) * 2 + 1] = (sm[0][1] + sm[1][1]) + (sm[2][1] + sm[3][1]);
  }
}

// out[0] = mean_b sqrt(S_b / n_b); coef[b] = 1 / (B * sqrt(S_b / n_b) * n_b): d out / d pred[b][i] = coef[b] * (log p - log g) / p on valid pixels
// (one wave per image, lanes over the partial sums, a shuffle tree: the one-thread loop over B * nb * 2 dependent loads took 83 us of the loss phase)
__global__ __launch_bounds__(256) void rmse_log_final_kernel(const double* __restrict__ part, int nb, int B, float* __restrict__ out, float* __restrict__ coef) {
  __shared__ double rs[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (B > 64) {                                            // (no LDS slot per image: the serial form)
    if (threadIdx.x != 0) return;
    double tot = 0.0;
    for (int b = 0; b < B; ++b) {
      double s = 0.0, c = 0.0;
      for (int k = 0; k < nb; ++k) { s += part[((size_t)b * nb + k) * 2]; c += part[((size_t)b * nb + k) * 2 + 1]; }
      const double r = sqrt(s / c);
      tot += r;
      coef[b] = (float)(1.0 / ((double)B * r * c));
    }
    out[0] = (float)(tot / B);
    return;
  }
  for (int b = wave; b < B; b += 4) {
    double s = 0.0, c = 0.0;
    for (int k = lane; k < nb; k += 64) { s += part[((size_t)b * nb + k) * 2]; c += part[((size_t)b * nb + k) * 2 + 1]; }
    s = wave_sum_d(s); c = wave_sum_d(c);
    if (lane == 0) {
      const double r = sqrt(s / c);
      rs[b] = r;
      coef[b] = (float)(1.0 / ((double)B * r * c));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int b = 0; b < B; ++b) tot += rs[b];
    out[0] = (float)(tot / B);
  }
}

__global__ __launch_bounds__(256) void rmse_log_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ coef,
                                                           const float* __restrict__ g, float* __restrict__ dpred, int HW, float min_depth, float clamp) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  const size_t k = (size_t)b * HW + i;
  const float gv = gt[k], pv = pred[k];
  float d = 0.f;
  if (gv > min_depth && pv > clamp) d = g[0] * coef[b] * (logf(pv) - logf(fmaxf(gv, clamp))) / pv;     // (pred <= clamp: the clamp's gradient is 0)
  dpred[k] = d;
}
}  // namespace

extern "C" int prn_loss_ws_doubles(int B) { return RED_BLOCKS * 2 * (B > 1 ? B : 1); }

extern "C" int prn_focal_sum_fwd(const float* x, const int64_t* labels, float* out, double* ws, int64_t rows, int C, float alpha, float gamma, void* stream) {
  PRN_REQUIRE(x && labels && out && ws && rows > 0 && C > 0, "prn_focal_sum_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = rows * C;
  int nb = cdiv(n, 256 * 4);
  nb = nb > RED_BLOCKS ? RED_BLOCKS : (nb < 1 ? 1 : nb);
  hipLaunchKernelGGL(focal_partial_kernel, dim3(nb), dim3(256), 0, st, x, labels, ws, n, C, alpha, gamma);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, nb, out);
  PRN_CHECK_LAUNCH("prn_focal_sum_fwd");
  return 0;
}

extern "C" int prn_focal_sum_bwd(const float* x, const int64_t* labels, const float* g_out, float* dx, int64_t rows, int C, float alpha, float gamma,
                                 void* stream) {
  PRN_REQUIRE(x && labels && g_out && dx && rows > 0 && C > 0, "prn_focal_sum_bwd: bad arguments");
  const int64_t n = rows * C;
  hipLaunchKernelGGL(focal_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, labels, g_out, dx, n, C, alpha, gamma);
  PRN_CHECK_LAUNCH("prn_focal_sum_bwd");
  return 0;
}

extern "C" int prn_rmse_log_fwd(const float* pred, const float* gt, float* out, float* coef, double* ws, int B, int HW, float min_depth, float clamp,
                                void* stream) {
  PRN_REQUIRE(pred && gt && out && coef && ws && B > 0 && HW > 0, "prn_rmse_log_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int nb = cdiv(HW, 256 * 8);
  nb = nb > RED_BLOCKS ? RED_BLOCKS : (nb < 1 ? 1 : nb);
  hipLaunchKernelGGL(rmse_log_partial_kernel, dim3(nb, B), dim3(256), 0, st, pred, gt, ws, HW, min_depth, clamp);
  hipLaunchKernelGGL(rmse_log_final_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nb, B, out, coef);
  PRN_CHECK_LAUNCH("prn_rmse_log_fwd");
  return 0;
}

extern "C" int prn_rmse_log_bwd(const float* pred, const float* gt, const float* coef, const float* g_out, float* dpred, int B, int HW, float min_depth,
                                float clamp, void* stream) {
  PRN_REQUIRE(pred && gt && coef && g_out && dpred && B > 0 && HW > 0, "prn_rmse_log_bwd: bad arguments");
  hipLaunchKernelGGL(rmse_log_bwd_kernel, dim3(cdiv(HW, 256), B), dim3(256), 0, (hipStream_t)stream, pred, gt, coef, g_out, dpred, HW, min_depth, clamp);
  PRN_CHECK_LAUNCH("prn_rmse_log_bwd");
  return 0;
}

// ---------------------------------------------------------------------------------------------- virtual-normal trimming rule
// vnl.py:106-117,133-165 after the per-triplet losses: per sampled region sort the valid losses, drop the lowest quarter, average
// the rest; per image sum the regions' means over (planes + 1 if the non-planar region had a valid triplet).  The operator chain
// (segment sums as differences of prefix sums, rank arithmetic, masks) was ~65 small launches around one sort; here: the sort key, the
// sort itself (ATen), one workgroup per region, one thread for the per-image combination.
namespace {
__global__ __launch_bounds__(256) void vnl_trim_key_kernel(const double* __restrict__ loss, const unsigned char* __restrict__ valid,
                                                           const int64_t* __restrict__ seg, double* __restrict__ key, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double l = loss[i];
  key[i] = (double)seg[i] * 4.0 + (valid[i] ? (isnan(l) ? 1.5 : l) : 2.0);      // valid ascending (losses are in [0, 1]), NaN, invalid
}

// Region s: sorted positions [seg_start[s], seg_start[s+1]) hold its triplets, the m valid ones first.  The regions differ in size by three orders of
// magnitude (a non-planar region: ~1e5 triplets, a small plane: a few hundred), and one workgroup per region took as long as the largest one (187 us): every
// region is cut into VT_CH chunks of max(1024, size / VT_CH) positions, grid (nseg, VT_CH) -- first the valid counts per chunk (the kept range [drop, m) needs the
// region's total), then the kept sums per chunk; the chunks are added in chunk order by vnl_trim_final_kernel (fixed order: deterministic).
constexpr int VT_CH = 16;
__device__ __forceinline__ void vt_chunk(const int64_t* __restrict__ seg_start, int nseg, int n, int s, int c, int& a, int& lo, int& hi) {
  a = (int)seg_start[s];
  const int b = s + 1 < nseg ? (int)seg_start[s + 1] : n, size = b - a;
  const int chk = max(1024, (size + VT_CH - 1) / VT_CH);
  lo = min(b, a + c * chk);
  hi = min(b, lo + chk);
}
__global__ __launch_bounds__(256) void vnl_trim_count_kernel(const unsigned char* __restrict__ valid, const int64_t* __restrict__ order,
                                                             const int64_t* __restrict__ seg_start, int nseg, int n, int* __restrict__ cnt) {
  const int s = blockIdx.x, c = blockIdx.y;
  int a, lo, hi;
  vt_chunk(seg_start, nseg, n, s, c, a, lo, hi);
  __shared__ int wc[4];
  int m = 0;
  for (int j = lo + threadIdx.x; j < hi; j += 256) m += valid[order[j]] ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m += __shfl_xor(m, o, 64);
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) cnt[s * VT_CH + c] = wc[0] + wc[1] + wc[2] + wc[3];
}
__global__ __launch_bounds__(256) void vnl_trim_sum_kernel(const double* __restrict__ loss, const int64_t* __restrict__ order, const int64_t* __restrict__ seg_start,
                                                           int nseg, int n, const int* __restrict__ cnt, double* __restrict__ part, int* __restrict__ seg_m) {
  const int s = blockIdx.x, c = blockIdx.y;
  int a, lo, hi;
  vt_chunk(seg_start, nseg, n, s, c, a, lo, hi);
  int m = 0;
#pragma unroll
  for (int k = 0; k < VT_CH; ++k) m += cnt[s * VT_CH + k];
  const int drop = m / 4;
  lo = max(lo, a + drop);
  hi = min(hi, a + m);
  __shared__ double wp[4];
  double acc = 0.0;
  for (int j = lo + threadIdx.x; j < hi; j += 256) {
    const double l = loss[order[j]];
    acc += isnan(l) ? 0.0 : l;
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) wp[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    part[s * VT_CH + c] = (wp[0] + wp[1]) + (wp[2] + wp[3]);
    if (c == 0) seg_m[s] = m;
  }
}

// out[b] = sum over the image's counted regions of seg_sum / (m - drop)  /  (planes + [non-planar region counted]);
// seg_coef[s] = d out[img] / d (a kept loss of region s)
// (one wave per image, lanes over the regions: the one-thread double loop over B * nseg took 64 us of the loss phase)
__global__ __launch_bounds__(256) void vnl_trim_final_kernel(const double* __restrict__ part, double* __restrict__ seg_sum, const int* __restrict__ seg_m,
                                                             const unsigned char* __restrict__ seg_is_plane, const int64_t* __restrict__ seg_img,
                                                             const double* __restrict__ nplanes, int nseg, int B, double* __restrict__ out,
                                                             double* __restrict__ seg_coef) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int s = threadIdx.x; s < nseg; s += 256) {            // the regions' kept sums: their chunks in chunk order
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < VT_CH; ++k) t += part[s * VT_CH + k];
    seg_sum[s] = t;
  }
  __syncthreads();
  for (int b = wave; b < B; b += 4) {
    double tot = 0.0, extra = 0.0;
    for (int s = lane; s < nseg; s += 64) {
      if (seg_img[s] != b) continue;
      const int m = seg_m[s], drop = m / 4;
      const bool np_ok = !seg_is_plane[s] && m > 0, use = seg_is_plane[s] || np_ok;
      if (use) tot += seg_sum[s] / (double)(m - drop);      // (a plane without a valid triplet: 0 / 0 = NaN like the reference)
      if (np_ok) extra += 1.0;
    }
    tot = wave_sum_d(tot); extra = wave_sum_d(extra);
    const double den = nplanes[b] + extra;
    if (lane == 0) out[b] = tot / den;
    for (int s = lane; s < nseg; s += 64) {
      if (seg_img[s] != b) continue;
      const int m = seg_m[s], drop = m / 4;
      const bool use = seg_is_plane[s] || m > 0;
      seg_coef[s] = (use && m - drop > 0) ? 1.0 / ((double)(m - drop) * den) : 0.0;
    }
  }
}

// One thread per SORTED position j (its region by bisection over seg_start): the regions differ in size by three orders of magnitude -- a non-planar region
// holds ~1e5 triplets, a small plane a few hundred -- and the one-workgroup-per-region form of this kernel took as long as its largest region (146 us).
__global__ __launch_bounds__(256) void vnl_trim_bwd_kernel(const double* __restrict__ loss, const int64_t* __restrict__ order, const int64_t* __restrict__ seg_start,
                                                           const int* __restrict__ seg_m, const double* __restrict__ seg_coef, const int64_t* __restrict__ seg_img,
                                                           const double* __restrict__ g, int nseg, int n, double* __restrict__ dloss) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  int lo = 0, hi = nseg - 1;                                 // largest s with seg_start[s] <= j (seg_start[0] == 0; empty regions share a start: the last one wins,
  while (lo < hi) {                                          //  and it is the one that owns position j)
    const int mid = (lo + hi + 1) >> 1;
    if ((int)seg_start[mid] <= j) lo = mid; else hi = mid - 1;
  }
  const int s = lo, a = (int)seg_start[s];
  const int m = seg_m[s], drop = m / 4;
  const double c = seg_coef[s] * g[seg_img[s]];
  const int64_t i = order[j];
  const bool kept = j - a >= drop && j - a < m && !isnan(loss[i]);
  dloss[i] = kept ? c : 0.0;
}
}  // namespace

extern "C" int prn_vnl_trim_key(const double* loss, const unsigned char* valid, const int64_t* seg, double* key, int n, void* stream) {
  PRN_REQUIRE(loss && valid && seg && key && n > 0, "prn_vnl_trim_key: bad arguments");
  hipLaunchKernelGGL(vnl_trim_key_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, loss, valid, seg, key, n);
  PRN_CHECK_LAUNCH("prn_vnl_trim_key");
  return 0;
}

extern "C" int64_t prn_vnl_trim_ws_bytes(int nseg) { return nseg > 0 ? (int64_t)nseg * VT_CH * (8 + 4) : -1; }

extern "C" int prn_vnl_trim_fwd(const double* loss, const unsigned char* valid, const int64_t* order, const int64_t* seg_start,
                                const unsigned char* seg_is_plane, const int64_t* seg_img, const double* nplanes, int nseg, int n, int B, double* out,
                                double* seg_sum, int* seg_m, double* seg_coef, void* ws, void* stream) {
  PRN_REQUIRE(loss && valid && order && seg_start && seg_is_plane && seg_img && nplanes && out && seg_sum && seg_m && seg_coef && ws && nseg > 0 && n > 0 && B > 0,
              "prn_vnl_trim_fwd: bad arguments");
  PRN_REQUIRE(nseg <= 65535 && (reinterpret_cast<uintptr_t>(ws) & 7) == 0, "prn_vnl_trim_fwd: at most 65535 regions, 8-byte aligned workspace");
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)ws;                              // [nseg][VT_CH] kept sums, then [nseg][VT_CH] valid counts
  int* cnt = (int*)(part + (size_t)nseg * VT_CH);
  hipLaunchKernelGGL(vnl_trim_count_kernel, dim3(nseg, VT_CH), dim3(256), 0, st, valid, order, seg_start, nseg, n, cnt);
  hipLaunchKernelGGL(vnl_trim_sum_kernel, dim3(nseg, VT_CH), dim3(256), 0, st, loss, order, seg_start, nseg, n, (const int*)cnt, part, seg_m);
  hipLaunchKernelGGL(vnl_trim_final_kernel, dim3(1), dim3(256), 0, st, (const double*)part, seg_sum, (const int*)seg_m, seg_is_plane, seg_img, nplanes, nseg, B, out,
                     seg_coef);
  PRN_CHECK_LAUNCH("prn_vnl_trim_fwd");
  return 0;
}

extern "C" int prn_vnl_trim_bwd(const double* loss, const int64_t* order, const int64_t* seg_start, const int* seg_m, const double* seg_coef,
                                const int64_t* seg_img, const double* g_out, int nseg, int n, double* dloss, void* stream) {
  PRN_REQUIRE(loss && order && seg_start && seg_m && seg_coef && seg_img && g_out && dloss && nseg > 0 && n > 0, "prn_vnl_trim_bwd: bad arguments");
  hipLaunchKernelGGL(vnl_trim_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, loss, order, seg_start, seg_m, seg_coef, seg_img, g_out, nseg, n, dloss);
  PRN_CHECK_LAUNCH("prn_vnl_trim_bwd");
  return 0;
}

// ---- depth-gradient weights of the lava term (models/functions/losses.py:288-329, GT only) -----------------------------------------------------
// w = min(sobel^2(gt) / max(gt, res)^2, 1e-2), zeroed below 1e-4, with sobel^2 = gx^2 + gy^2 of the reflect-padded 3x3 Sobel / 8: one pass instead of
// the ~20 elementwise launches of the tensor formulation (reflection pad, eight shifted views, two gradients, square, clamp, divide, clamp, compare, where).
// Same operations in the same order and precision as the formulation evaluated in IEEE arithmetic (no a * b + c contraction: see the pragma).
namespace {
__global__ __launch_bounds__(256) void lava_gt_kernel(const float* __restrict__ gt, float* __restrict__ out, int64_t total, int H, int W, float res) {
#pragma clang fp contract(off)   // (hipcc contracts a * b + c into one fma by default, and __fmul_rn / __fadd_rn are plain operators in this toolchain)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int x, y;
  int64_t b;
  prn_idx3(i, W, H, x, y, b);
  const float* g = gt + b * H * W;
  const int ym = y > 0 ? y - 1 : 1, yp = y < H - 1 ? y + 1 : H - 2;          // ReflectionPad2d(1)
  const int xm = x > 0 ? x - 1 : 1, xp = x < W - 1 ? x + 1 : W - 2;
  const float tl = g[ym * W + xm], tc = g[ym * W + x], tr = g[ym * W + xp];
  const float ml = g[y * W + xm], mc = g[y * W + x], mr = g[y * W + xp];
  const float bl = g[yp * W + xm], bc = g[yp * W + x], br = g[yp * W + xp];
  float gx = tl - tr;                                                        // (plain operators: the __f*_rn wrappers are inline functions compiled with contraction allowed)
  gx = gx + 2.f * ml; gx = gx - 2.f * mr; gx = gx + bl; gx = gx - br;        // (2 * v is exact, fused or not)
  gx = gx / 8.f;
  float gy = tl + 2.f * tc;
  gy = gy + tr; gy = gy - bl; gy = gy - 2.f * bc; gy = gy - br;
  gy = gy / 8.f;
  const float gx2 = gx * gx, gy2 = gy * gy;
  const float s = gx2 + gy2;
  const float d = fmaxf(mc, res);
  const float d2 = d * d;
  float w = s / d2;
  w = fminf(w, 1e-2f);
  out[i] = w < 1e-4f ? 0.f : w;
}
}  // namespace

extern "C" int prn_lava_gt_weights(const float* gt, float* out, int B, int H, int W, float depth_resolution, void* stream) {
  PRN_REQUIRE(gt && out && B > 0 && H >= 2 && W >= 2, "prn_lava_gt_weights: bad arguments");
  const int64_t total = (int64_t)B * H * W;
  hipLaunchKernelGGL(lava_gt_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, gt, out, total, H, W, depth_resolution);
  PRN_CHECK_LAUNCH("prn_lava_gt_weights");
  return 0;
}

// Adam step over every parameter of the model in ONE launch (reference train.py:251-256: optim.Adam over five parameter
// groups; train.py:362 optimizer.step()).  HBM-bound: reads p, g, m, v and writes p, m, v -- 28 bytes per element, 1.6 GB for
// PlaneRecNet_101; torch's fused multi-tensor Adam takes 14 launches and 0.69 ms for it (2.4 TB/s).
// Work list: fixed-size chunks of the parameter tensors (one workgroup per chunk), built once by the caller; the tensors stay
// where the framework allocated them (tables of device pointers), only the gradient pointers change from step to step.
#include "prn_common.h"

namespace {
constexpr int ADAM_CHUNK = 4096;       // elements per workgroup (256 threads x 4 float4)

__global__ __launch_bounds__(256) void adam_kernel(const int2* __restrict__ chunks, float* const* __restrict__ p, const float* const* __restrict__ g,
                                                   float* const* __restrict__ m, float* const* __restrict__ v, const int* __restrict__ numel,
                                                   const float* __restrict__ lr, const float* __restrict__ step, const float* __restrict__ found_inf,
                                                   const float* __restrict__ grad_scale, double beta1_d, double beta2_d, float eps,
                                                   const float* __restrict__ present, const int* __restrict__ present_idx) {
  if (found_inf && *found_inf != 0.f) return;                 // collective "skip this update" (train.py:353), decided on the device
  const int2 c = chunks[blockIdx.x];
  const int t = c.x, off = c.y;
  // data-parallel runs: a tensor for which NO rank produced a gradient is left alone (parameter and moments), like optim.Adam
  // leaves a parameter whose .grad is None (train.py:362) -- the exchange zero-fills such gradients so that every rank posts the
  // same collectives, and hands over the all-reduced presence counts instead of a host-side None
  if (present && present[present_idx[t]] == 0.f) return;
  const int n = min(ADAM_CHUNK, numel[t] - off);
  // bias corrections of step + 1 (the counter is advanced by adam_advance_kernel after this launch), in double like torch
  const double s = (double)step[t] + 1.0;                     // per tensor, like optim.Adam's state['step']: a skipped tensor does not advance
  const float bc1 = (float)(1.0 - pow(beta1_d, s)), bc2_sqrt = (float)sqrt(1.0 - pow(beta2_d, s));
  // (1 - beta) is formed in double and then rounded, like torch: 1 - 0.999f would be 0.00100005)
  const float beta2 = (float)beta2_d, w1 = (float)(1.0 - beta1_d), w2 = (float)(1.0 - beta2_d);
  const float step_size = lr[t] / bc1;
  const float inv_scale = grad_scale ? 1.f / *grad_scale : 1.f;
  float* pp = p[t] + off;
  const float* gp = g[t] + off;
  float* mp = m[t] + off;
  float* vp = v[t] + off;
  auto upd = [&](float& pw, float gw, float& mw, float& vw) {
    gw *= inv_scale;
    mw = mw + w1 * (gw - mw);                                  // lerp(exp_avg, grad, 1 - beta1)
    vw = beta2 * vw + w2 * gw * gw;
    const float denom = sqrtf(vw) / bc2_sqrt + eps;
    pw -= step_size * mw / denom;
  };
  const bool vec = ((reinterpret_cast<uintptr_t>(pp) | reinterpret_cast<uintptr_t>(gp) | reinterpret_cast<uintptr_t>(mp) | reinterpret_cast<uintptr_t>(vp)) & 15) == 0;
  if (vec) {
    const int n4 = n >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
      float4 pw = reinterpret_cast<float4*>(pp)[i], mw = reinterpret_cast<float4*>(mp)[i], vw = reinterpret_cast<float4*>(vp)[i];
      const float4 gw = reinterpret_cast<const float4*>(gp)[i];
      upd(pw.x, gw.x, mw.x, vw.x); upd(pw.y, gw.y, mw.y, vw.y); upd(pw.z, gw.z, mw.z, vw.z); upd(pw.w, gw.w, mw.w, vw.w);
      reinterpret_cast<float4*>(pp)[i] = pw; reinterpret_cast<float4*>(mp)[i] = mw; reinterpret_cast<float4*>(vp)[i] = vw;
    }
    for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) upd(pp[i], gp[i], mp[i], vp[i]);
  } else {
    for (int i = threadIdx.x; i < n; i += 256) upd(pp[i], gp[i], mp[i], vp[i]);
  }
}

__global__ void adam_advance_kernel(float* step, int ntensors, const float* found_inf, const float* present, const int* present_idx) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntensors || (found_inf && *found_inf != 0.f)) return;
  if (present && present[present_idx[t]] == 0.f) return;
  step[t] += 1.f;
}
}  // namespace

extern "C" int prn_adam_chunk_elems(void) { return ADAM_CHUNK; }

extern "C" int prn_adam_step(const int* chunks, int nchunks, int ntensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                             const int* numel, const float* lr, float* step, const float* found_inf, const float* grad_scale, double beta1, double beta2,
                             float eps, void* stream) {
  return prn_adam_step_masked(chunks, nchunks, ntensors, p, g, m, v, numel, lr, step, found_inf, grad_scale, beta1, beta2, eps, nullptr, nullptr, stream);
}

extern "C" int prn_adam_step_masked(const int* chunks, int nchunks, int ntensors, float* const* p, const float* const* g, float* const* m,
                                    float* const* v, const int* numel, const float* lr, float* step, const float* found_inf, const float* grad_scale, double beta1,
                                    double beta2, float eps, const float* present, const int* present_idx, void* stream) {
  PRN_REQUIRE(chunks && p && g && m && v && numel && lr && step && nchunks > 0 && ntensors > 0, "prn_adam_step: bad arguments");
  PRN_REQUIRE((present == nullptr) == (present_idx == nullptr), "prn_adam_step_masked: present and present_idx come together");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adam_kernel, dim3(nchunks), dim3(256), 0, st, reinterpret_cast<const int2*>(chunks), p, g, m, v, numel, lr, (const float*)step, found_inf,
                     grad_scale, beta1, beta2, eps, present, present_idx);
  PRN_CHECK_LAUNCH("prn_adam_step");
  hipLaunchKernelGGL(adam_advance_kernel, dim3(cdiv(ntensors, 256)), dim3(256), 0, st, step, ntensors, found_inf, present, present_idx);
  PRN_CHECK_LAUNCH("prn_adam_step/advance");
  return 0;
}

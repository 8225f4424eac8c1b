// Lab: fp32 GEMM on the bf16 matrix pipe by exact three-way operand splitting.
//   x = h + m + l  with h, m, l the three consecutive 8-bit slices of x's 24-bit significand (truncation: every piece is exactly a bf16,
//   the sum is exactly x);  a*b = sum of the 9 piece products, each EXACT in fp32;  NPROD = 6 drops m*l, l*m, l*l (<= 2^-23 |a*b|).
// y[z][b][m][p] = sum_k w[z][m][k] * x[z][b][k][p]  (the 1x1 convolution / Winograd-product form), against conv_igemm_kernel on the same shapes
// and against an fp64 reference for the error of both.
//   hipcc --offload-arch=gfx950 -O2 split_lab.cpp -o split_lab.bin -L../../../planerecnet_amd -lprn_hip -Wl,-rpath,'$ORIGIN/../../../planerecnet_amd'
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <chrono>
#include "../../../include/prn.h"
#include "split_gemm.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int M, K, B, HW, Z; int epi; bool add, bias; };

__global__ void ref64_kernel(const float* w, const float* x, const float* bias, const float* addend, double* y, double* mag, int M, int K, int B, int HW, int Z, int epi) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)Z * B * M * HW;
  if (i >= total) return;
  const int p = i % HW; const int m = (i / HW) % M; const int b = (i / ((long long)HW * M)) % B; const int z = i / ((long long)HW * M * B);
  const float* a = w + ((long long)z * M + m) * K;
  const float* xx = x + ((long long)z * B + b) * K * HW + p;
  double acc = 0., mg = 0.;
  for (int k = 0; k < K; ++k) { const double pr = (double)a[k] * (double)xx[(long long)k * HW]; acc += pr; mg += fabs(pr); }
  if (bias) acc += bias[m];
  if (addend) acc += addend[i];
  if (epi == PRN_EPI_RELU) acc = acc > 0. ? acc : 0.;
  y[i] = acc; mag[i] = mg;
}
__global__ void fill_kernel(float* p, long long n, unsigned seed, float scale) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  p[i] = ((h & 0xffffff) / 16777216.0f - 0.5f) * 2.f * scale;
}
__global__ void flush_kernel(float* p, long long n, float v) { const long long i = (long long)blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = v; }
static float* dalloc(long long n) { float* p; CK(hipMalloc(&p, n * 4)); return p; }
static void fill(float* p, long long n, unsigned seed, float scale) { fill_kernel<<<(unsigned)((n + 255) / 256), 256>>>(p, n, seed, scale); }

int main(int argc, char** argv) {
  const char* filter = argc > 1 ? argv[1] : nullptr;
  const int reps = getenv("LAB_REPS") ? atoi(getenv("LAB_REPS")) : 20;
  const int nprod = getenv("LAB_NPROD") ? atoi(getenv("LAB_NPROD")) : 6;
  std::vector<Shape> shapes = {
      {"s3 256->1024 @30x40", 1024, 256, 8, 1200, 1, 0, false, false},
      {"s3 1024->256 @30x40", 256, 1024, 8, 1200, 1, 0, false, false},
      {"s3 256->1024 +add+relu", 1024, 256, 8, 1200, 1, 1, true, true},
      {"wino 256x256 P=640 z36", 256, 256, 1, 640, 36, 0, false, false},
      {"wino 256x256 P=9600 z36", 256, 256, 1, 9600, 36, 0, false, false},
      {"wino 128x128 P=2400 z36", 128, 128, 1, 2400, 36, 0, false, false},
      {"wino 64x64 P=9600 z36", 64, 64, 1, 9600, 36, 0, false, false},
      {"wino 512x512 P=160 z36", 512, 512, 1, 160, 36, 0, false, false},
      {"s1 64->256 @120x160", 256, 64, 8, 19200, 1, 0, false, false},
      {"s1 256->64 @120x160", 64, 256, 8, 19200, 1, 0, false, false},
      {"s2 128->512 @60x80", 512, 128, 8, 4800, 1, 0, false, false},
      {"s2 512->128 @60x80", 128, 512, 8, 4800, 1, 0, false, false},
      {"fpn 256->256 @120x160", 256, 256, 8, 19200, 1, 0, false, true},
      {"s4 512->2048 @15x20", 2048, 512, 8, 300, 1, 0, false, false},
      {"s4 2048->512 @15x20", 512, 2048, 8, 300, 1, 0, false, false},
      {"tail M=200 K=72 HW=1204", 200, 72, 3, 1204, 2, 1, true, true},
      {"gemm 4096^2 x 16384", 4096, 4096, 1, 16384, 1, 0, false, false},
  };
  hipStream_t st; CK(hipStreamCreate(&st));
  const bool cold = getenv("LAB_COLD") && atoi(getenv("LAB_COLD"));
  float* flushbuf = nullptr;
  if (cold) CK(hipMalloc(&flushbuf, 512ll << 20));
  printf("NPROD = %d\n%-26s %8s | %8s %7s %7s | %8s %7s | %s\n", nprod, "shape", "GFLOP", "split us", "TF/s eq", "presplit", "old us", "TF/s", "error / sum|a||b| : split max, rms | fp32 MFMA max, rms");
  for (const Shape& s : shapes) {
    if (filter && !strstr(s.name, filter)) continue;
    const long long N = (long long)s.B * s.HW;
    const long long na = (long long)s.Z * s.K * s.M, nx = (long long)s.Z * s.K * N, ny = (long long)s.Z * s.M * N;
    float *w = dalloc(na), *x = dalloc(nx), *y = dalloc(ny), *yold = dalloc(ny), *add = s.add ? dalloc(ny) : nullptr, *bias = s.bias ? dalloc(s.M) : nullptr;
    fill(w, na, 1u, 1.0f / sqrtf((float)s.K)); fill(x, nx, 2u, 1.f);
    if (add) fill(add, ny, 3u, 1.f);
    if (bias) fill(bias, s.M, 4u, 1.f);
    CK(hipMemset(y, 0xff, ny * 4));
    const long long wsb = split_gemm_ws_bytes(s.M, s.K, s.Z);
    void* ws; CK(hipMalloc(&ws, wsb));
    auto run_presplit = [&]() { split_gemm_prepare(w, ws, s.M, s.K, s.Z, st); };
    auto run_split = [&]() { split_gemm_run(ws, x, bias, add, y, s.M, s.K, s.B, s.HW, s.Z, s.epi, nprod, st); };
    prn_conv_desc c; memset(&c, 0, sizeof(c));
    c.B = s.B; c.C = s.K; c.H = 1; c.W = s.HW; c.M = s.M; c.KH = c.KW = 1; c.stride = 1; c.pad = 0; c.Ho = 1; c.Wo = s.HW; c.in_mode = PRN_IN_ZERO; c.dil = 1;
    c.epilogue = s.epi; c.ystride = 1; c.yH = 1; c.yW = s.HW;
    const long long owsb = s.Z == 1 ? prn_conv2d_fwd_ws_bytes(&c) : 0;
    float* ows = owsb > 0 ? dalloc(owsb / 4) : nullptr;
    const bool old_ok = s.Z == 1 || (!s.add && !s.bias && s.epi == 0);
    auto run_old = [&]() {
      int rc = s.Z == 1 ? prn_conv2d_fwd(&c, x, w, bias, add, yold, ows, st) : prn_gemm_batched(s.M, s.K, s.HW * s.B, s.Z, w, x, yold, st);
      if (rc) { printf("old: %s\n", prn_last_error()); exit(1); }
    };
    auto timeit = [&](auto fn) {
      fn(); CK(hipStreamSynchronize(st));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      if (cold) {                                            // LAB_COLD=1: 512 MB written between launches (operands come from HBM, as inside a training step)
        float tot = 0.f; const int n = reps < 12 ? reps : 12;
        for (int i = 0; i < n; ++i) {
          flush_kernel<<<(512 << 20) / 4 / 256, 256, 0, st>>>(flushbuf, (512ll << 20) / 4, (float)i);
          CK(hipEventRecord(e0, st)); fn(); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
        }
        return tot / n * 1e3f;
      }
      CK(hipEventRecord(e0, st));
      const auto w0 = std::chrono::steady_clock::now();
      for (int i = 0; i < reps; ++i) fn();
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (getenv("LAB_WALL")) printf("    [events %.1f us/launch, host wall %.1f us/launch over %d launches]\n", ms / reps * 1e3, wall / reps * 1e6, reps);
      return ms / reps * 1e3f;
    };
    const char* only = getenv("LAB_ONLY");                       // "split" / "old": time (and keep the GPU busy with) one kernel only
    run_presplit();
    const float tpre = only ? 0.f : timeit(run_presplit);
    const float tsp = (only && strcmp(only, "split")) ? 0.f : timeit(run_split);
    const float told = (old_ok && !(only && strcmp(only, "old"))) ? timeit(run_old) : 0.f;
    const double gf = 2.0 * s.M * s.K * (double)N * s.Z / 1e9;
    char msg[160] = "-";
    if (ny <= (48ll << 20)) {
      double *yref, *mag; CK(hipMalloc(&yref, ny * 8)); CK(hipMalloc(&mag, ny * 8));
      ref64_kernel<<<(unsigned)((ny + 255) / 256), 256, 0, st>>>(w, x, bias, add, yref, mag, s.M, s.K, s.B, s.HW, s.Z, s.epi);
      CK(hipStreamSynchronize(st));
      std::vector<float> hy(ny), ho(ny); std::vector<double> hr(ny), hm(ny);
      CK(hipMemcpy(hy.data(), y, ny * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ho.data(), yold, ny * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hr.data(), yref, ny * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hm.data(), mag, ny * 8, hipMemcpyDeviceToHost));
      double mx[2] = {0, 0}, sq[2] = {0, 0}, bias_[2] = {0, 0}; long long nbad = 0;
      for (long long i = 0; i < ny; ++i) {
        const double d = hm[i] + fabs(hr[i]) + 1e-30;
        const double e0 = ((double)hy[i] - hr[i]) / d, e1 = old_ok ? ((double)ho[i] - hr[i]) / d : 0.;
        if (!(e0 == e0)) ++nbad;
        if (fabs(e0) > mx[0]) mx[0] = fabs(e0);
        if (fabs(e1) > mx[1]) mx[1] = fabs(e1);
        sq[0] += e0 * e0; sq[1] += e1 * e1; bias_[0] += e0; bias_[1] += e1;
      }
      snprintf(msg, sizeof msg, "%.2e %.2e (mean %+.1e) | %.2e %.2e (mean %+.1e) nan %lld", mx[0], sqrt(sq[0] / ny), bias_[0] / ny, mx[1], sqrt(sq[1] / ny), bias_[1] / ny, nbad);
      hipFree(yref); hipFree(mag);
    }
    printf("%-26s %8.2f | %8.1f %7.1f %7.1f | %8.1f %7.1f | %s\n", s.name, gf, tsp, gf / tsp * 1e3, tpre, told, told > 0 ? gf / told * 1e3 : 0.0, msg);
    fflush(stdout);
    hipFree(w); hipFree(x); hipFree(y); hipFree(yold); if (add) hipFree(add); if (bias) hipFree(bias); hipFree(ws); if (ows) hipFree(ows);
  }
  return 0;
}

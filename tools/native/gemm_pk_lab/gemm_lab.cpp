// Lab bench for the persistent strip GEMM (prn_gemm_kn) against the tile-per-workgroup kernel (prn_conv2d_fwd / prn_gemm_batched)
// on the plain-GEMM shapes of a PlaneRecNet_101 training step (B = 8, 480x640).  No torch: starts in a second.
//   hipcc --offload-arch=gfx950 -O2 tools/native/gemm_lab.cpp -o tools/native/gemm_lab.bin -Lplanerecnet_amd -lprn_hip -Wl,-rpath,'$ORIGIN/../../planerecnet_amd'
//   tools/native/gemm_lab.bin [filter] ; env: PRN_PK_STAGES, PRN_PK_GRID, PRN_PK_SPLITS, LAB_REPS, LAB_CHECK=0, LAB_COLD=1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "../../../include/prn.h"
#ifdef LAB_NO_PK   // library without the lab kernel: time the product path only
typedef struct prn_gemm_desc { int32_t M, K, B, HW, lda, nz, epilogue, reserved; int64_t zat, zx, zy; } prn_gemm_desc;
static int64_t prn_gemm_kn_ws_bytes(const prn_gemm_desc*) { return 0; }
static int prn_gemm_kn(const prn_gemm_desc*, const float*, const float*, const float*, const float*, float*, void*, void*) { return 0; }
#else
extern "C" { typedef struct prn_gemm_desc { int32_t M, K, B, HW, lda, nz, epilogue, reserved; int64_t zat, zx, zy; } prn_gemm_desc;
int64_t prn_gemm_kn_ws_bytes(const prn_gemm_desc* d);
int prn_gemm_kn(const prn_gemm_desc* d, const float* at, const float* x, const float* bias, const float* addend, float* y, void* ws, void* stream); }
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int M, K, B, HW, Z; int epi; bool add, bias; };

// y[z][b][m][p] = sum_k at[z][k][m] * x[z][b][k][p], one k-ordered fmaf chain per element (what the MFMA computes unsplit)
__global__ void ref_kernel(const float* at, const float* x, const float* bias, const float* addend, float* y, int M, int K, int B, int HW, int Z, int epi) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)Z * B * M * HW;
  if (i >= total) return;
  const int p = i % HW; const int m = (i / HW) % M; const int b = (i / ((long long)HW * M)) % B; const int z = i / ((long long)HW * M * B);
  const float* a = at + (long long)z * K * M + m;
  const float* xx = x + ((long long)z * B + b) * K * HW + p;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = fmaf(a[(long long)k * M], xx[(long long)k * HW], acc);
  if (bias) acc += bias[m];
  if (addend) acc += addend[i];
  if (epi == PRN_EPI_RELU) acc = fmaxf(acc, 0.f);
  y[i] = acc;
}
__global__ void fill_kernel(float* p, long long n, unsigned seed, float scale) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  p[i] = ((h & 0xffffff) / 16777216.0f - 0.5f) * 2.f * scale;
}
__global__ void transpose_kernel(const float* at, float* w, int K, int M) {   // w[m][k] = at[k][m]
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)K * M) return;
  const int k = i % K, m = i / K;
  w[i] = at[(long long)k * M + m];
}
__global__ void flush_kernel(float* p, long long n, float v) { const long long i = (long long)blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = v; }

static float* dalloc(long long n) { float* p; CK(hipMalloc(&p, n * 4)); return p; }
static void fill(float* p, long long n, unsigned seed, float scale) { fill_kernel<<<(unsigned)((n + 255) / 256), 256>>>(p, n, seed, scale); }

int main(int argc, char** argv) {
  const char* filter = argc > 1 ? argv[1] : nullptr;
  const int reps = getenv("LAB_REPS") ? atoi(getenv("LAB_REPS")) : 20;
  const bool check = !getenv("LAB_CHECK") || atoi(getenv("LAB_CHECK"));
  const bool cold = getenv("LAB_COLD") && atoi(getenv("LAB_COLD"));
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
      {"s2 512->256 @60x80", 256, 512, 8, 4800, 1, 0, false, false},
      {"tail M=200 K=72 HW=1204", 200, 72, 3, 1204, 2, 1, true, true},
      {"gemm 4096^2 x 16384", 4096, 4096, 1, 16384, 1, 0, false, false},
  };
  hipStream_t st; CK(hipStreamCreate(&st));
  float* flushbuf = cold ? dalloc(256ll << 20) : nullptr;
  printf("%-28s %8s | %9s %7s %5s | %9s %7s | %s\n", "shape", "GFLOP", "pk us", "TF/s", "S", "old us", "TF/s", "check");
  for (const Shape& s : shapes) {
    if (filter && !strstr(s.name, filter)) continue;
    const long long N = (long long)s.B * s.HW;
    const long long na = (long long)s.Z * s.K * s.M, nx = (long long)s.Z * s.K * N, ny = (long long)s.Z * s.M * N;
    float *at = dalloc(na), *w = dalloc(na), *x = dalloc(nx), *y = dalloc(ny), *yref = dalloc(ny), *yold = dalloc(ny), *add = s.add ? dalloc(ny) : nullptr, *bias = s.bias ? dalloc(s.M) : nullptr;
    fill(at, na, 1u, 1.0f / sqrtf((float)s.K)); fill(x, nx, 2u, 1.f);
    if (add) fill(add, ny, 3u, 1.f);
    if (bias) fill(bias, s.M, 4u, 1.f);
    for (int z = 0; z < s.Z; ++z) transpose_kernel<<<(unsigned)(((long long)s.K * s.M + 255) / 256), 256>>>(at + (long long)z * s.K * s.M, w + (long long)z * s.K * s.M, s.K, s.M);
    CK(hipMemset(y, 0xff, ny * 4));
    prn_gemm_desc d; memset(&d, 0, sizeof(d));
    d.M = s.M; d.K = s.K; d.B = s.B; d.HW = s.HW; d.lda = s.M; d.nz = s.Z; d.epilogue = s.epi;
    d.zat = (long long)s.K * s.M; d.zx = (long long)s.K * N; d.zy = (long long)s.M * N;
    const long long wsb = prn_gemm_kn_ws_bytes(&d);
    float* ws = wsb > 0 ? dalloc(wsb / 4) : nullptr;
    const int S = wsb > 0 ? (int)(wsb / (ny * 4)) : 1;
    auto run_pk = [&]() { if (prn_gemm_kn(&d, at, x, bias, add, y, ws, st)) { printf("prn_gemm_kn: %s\n", prn_last_error()); exit(1); } };
    // old path
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
      float tot = 0.f;
      if (cold) {
        for (int i = 0; i < 6; ++i) {
          flush_kernel<<<(256 << 20) / 256, 256, 0, st>>>(flushbuf, 256ll << 20, (float)i);
          CK(hipEventRecord(e0, st)); fn(); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
        }
        return tot / 6 * 1e3f;
      }
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) fn();
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      return ms / reps * 1e3f;
    };
    if (getenv("LAB_TRACE")) {
      const int G = getenv("PRN_PK_GRID") ? atoi(getenv("PRN_PK_GRID")) : 512;
      unsigned long long* tb; CK(hipMalloc(&tb, (size_t)G * 64 * 8)); CK(hipMemset(tb, 0, (size_t)G * 64 * 8));
      char buf[64]; snprintf(buf, sizeof buf, "%llu", (unsigned long long)tb);
      run_pk(); run_pk(); CK(hipStreamSynchronize(st));
      setenv("PRN_PK_TRACE", buf, 1);
      run_pk(); CK(hipStreamSynchronize(st));
      unsetenv("PRN_PK_TRACE");
      std::vector<unsigned long long> h((size_t)G * 64);
      CK(hipMemcpy(h.data(), tb, (size_t)G * 64 * 8, hipMemcpyDeviceToHost));
      unsigned long long c0 = ~0ull, c1 = 0, w0 = ~0ull, w1 = 0;
      for (int g = 0; g < G; ++g) { const unsigned long long* t = &h[(size_t)g * 64]; if (!t[3]) continue; if (t[0] < c0) c0 = t[0]; if (t[3] > c1) c1 = t[3]; if (t[1] < w0) w0 = t[1]; if (t[4] > w1) w1 = t[4]; }
      printf("TRACE %s: kernel span %llu cycles, %.2f us wall (100 MHz clock) -> %.2f GHz\n", s.name, c1 - c0, (w1 - w0) / 100.0, (double)(c1 - c0) / ((w1 - w0) * 10.0));
      if (!getenv("PRN_PK_V") || atoi(getenv("PRN_PK_V")) == 2) {
        double A[5] = {0, 0, 0, 0, 0}, span = 0, wall = 0; int nwg = 0;
        for (int g = 0; g < G; ++g) {
          const unsigned long long* t = &h[(size_t)g * 64];
          if (!t[3]) continue;
          ++nwg; span += t[3] - t[0]; wall += (t[4] - t[1]) * 10.0;
          for (int i = 0; i < 5; ++i) A[i] += t[8 + i];
        }
        printf("  v2: %d workgroups, mean span %.0f cycles = %.1f us -> %.2f GHz; slices/wg %.1f, mean TN %.2f\n", nwg, span / nwg, wall / nwg / 1e3, span / wall, A[3] / nwg, A[4] / A[3]);
        printf("  per slice (wave 0): at vmcnt %.0f, at barrier %.0f cycles; slices with a piece %.1f%%; MFMA floor per slice (two waves) %.0f; span / slice %.0f\n", A[0] / A[3], A[1] / A[3],
               100.0 * A[2] / A[3], A[4] / A[3] * 8 * 64 * 2, span / A[3]);
      } else {
      double sums[5] = {0, 0, 0, 0, 0}; long long nch = 0; double wgspan = 0, startlag = 0, endlead = 0; int nwg = 0;
      for (int g = 0; g < G; ++g) {
        const unsigned long long* t = &h[(size_t)g * 64];
        if (!t[3]) continue;
        ++nwg; wgspan += t[3] - t[0]; startlag += t[0] - c0; endlead += c1 - t[3];
        for (int c = 0; c < (int)t[5] && c < 9; ++c) {
          const unsigned long long* q = t + 8 + 6 * c;
          sums[0] += q[1] - q[0]; sums[1] += q[2] - q[1]; sums[2] += q[3]; sums[3] += q[4] - q[2]; sums[4] += (double)(q[5] >> 32) * (unsigned)q[5] * 512.0; ++nch;
        }
      }
      printf("  per workgroup (mean of %d): span %.0f cyc, start lag %.0f, idle at end %.0f; chunks %.2f\n", nwg, wgspan / nwg, startlag / nwg, endlead / nwg, (double)nch / nwg);
      printf("  per chunk (mean): first slice ready %.0f, K loop %.0f (of it at wait+barrier %.0f; MFMA-issue floor %.0f), epilogue %.0f\n", sums[0] / nch, sums[1] / nch,
             sums[2] / nch, sums[4] / nch, sums[3] / nch);
      for (int g = 0; g < G; g += G / 8 + 1) {
        const unsigned long long* t = &h[(size_t)g * 64];
        printf("  wg %3d xcc %llu hwid %08llx start %6llu end %6llu :", g, t[2] >> 32, t[2] & 0xffffffffull, t[0] - c0, t[3] - c0);
        for (int c = 0; c < (int)t[5] && c < 9; ++c) { const unsigned long long* q = t + 8 + 6 * c; printf(" [tn%llu nk%u pro %llu loop %llu wait %llu epi %llu]", q[5] >> 32, (unsigned)q[5], q[1] - q[0], q[2] - q[1], q[3], q[4] - q[2]); }
        printf("\n");
      }
      }
      hipFree(tb);
    }
    const float tpk = timeit(run_pk);
    const float told = old_ok ? timeit(run_old) : 0.f;
    const double gf = 2.0 * s.M * s.K * (double)N * s.Z / 1e9;
    char msg[128] = "-";
#ifdef LAB_NO_PK
    if (false) {
#else
    if (check && ny <= (64ll << 20)) {
#endif
      ref_kernel<<<(unsigned)((ny + 255) / 256), 256, 0, st>>>(at, x, bias, add, yref, s.M, s.K, s.B, s.HW, s.Z, s.epi);
      CK(hipMemsetAsync(y, 0xff, ny * 4, st));
      run_pk();
      CK(hipStreamSynchronize(st));
      std::vector<float> hy(ny), hr(ny);
      CK(hipMemcpy(hy.data(), y, ny * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), yref, ny * 4, hipMemcpyDeviceToHost));
      double maxerr = 0, maxref = 0; long long nbad = 0, nneq = 0;
      for (long long i = 0; i < ny; ++i) {
        const double e = fabs((double)hy[i] - hr[i]);
        if (!(e == e)) ++nbad;
        if (hy[i] != hr[i]) ++nneq;
        if (e > maxerr) maxerr = e;
        if (fabs(hr[i]) > maxref) maxref = fabs(hr[i]);
      }
      snprintf(msg, sizeof msg, "maxerr %.2e (max %.2f) neq %lld nan %lld %s", maxerr, maxref, nneq, nbad, (nbad == 0 && maxerr <= 2e-5 * maxref + 1e-6) ? "OK" : "FAIL");
    }
    printf("%-28s %8.2f | %9.1f %7.1f %5d | %9.1f %7.1f | %s\n", s.name, gf, tpk, gf / tpk * 1e3, S, told, told > 0 ? gf / told * 1e3 : 0.0, msg);
    fflush(stdout);
    hipFree(at); hipFree(w); hipFree(x); hipFree(y); hipFree(yref); hipFree(yold); if (add) hipFree(add); if (bias) hipFree(bias); if (ws) hipFree(ws); if (ows) hipFree(ows);
  }
  return 0;
}

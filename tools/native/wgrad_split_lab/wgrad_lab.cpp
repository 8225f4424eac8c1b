// Lab bench: wgrad_split_kernel (wgrad_split.h) against the library's fp32 MFMA weight-gradient kernel on the 1x1 layers of the training step.
//   hipcc --offload-arch=gfx950 -O3 wgrad_lab.cpp -o wgrad_lab.bin -L../../../planerecnet_amd -lprn_hip -Wl,-rpath,'$ORIGIN/../../../planerecnet_amd'
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "../../../include/prn.h"
#include "wgrad_split.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
struct Shape { const char* name; int C, M, B, HW; };
__global__ void fill_kernel(float* p, long long n, unsigned seed, float scale) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  p[i] = ((h & 0xffffff) / 16777216.0f - 0.5f) * 2.f * scale;
}
__global__ void ref64_kernel(const float* x, const float* dy, double* dw, double* mag, int B, int C, int M, int HW) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M * C) return;
  const int m = i / C, c = i % C;
  double s = 0, g = 0;
  for (int b = 0; b < B; ++b) {
    const float* a = dy + ((long long)b * M + m) * HW; const float* v = x + ((long long)b * C + c) * HW;
    for (int p = 0; p < HW; ++p) { const double pr = (double)a[p] * v[p]; s += pr; g += fabs(pr); }
  }
  dw[i] = s; mag[i] = g;
}
__global__ void flush_kernel(float* p, long long n, float v) { const long long i = (long long)blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = v; }
static float* dalloc(long long n) { float* p; CK(hipMalloc(&p, n * 4)); return p; }
int main(int argc, char** argv) {
  const char* filter = argc > 1 ? argv[1] : nullptr;
  const int reps = getenv("LAB_REPS") ? atoi(getenv("LAB_REPS")) : 100;
  const bool cold = getenv("LAB_COLD") && atoi(getenv("LAB_COLD"));
  const int dbg = getenv("WS_DBG") ? atoi(getenv("WS_DBG")) : 0;
  const int wgs = getenv("WS_WGS") ? atoi(getenv("WS_WGS")) : 768;
  std::vector<Shape> shapes = {{"s3 dW 1024x256 @30x40", 256, 1024, 8, 1200}, {"s3 dW 256x1024 @30x40", 1024, 256, 8, 1200}, {"s2 dW 512x128 @60x80", 128, 512, 8, 4800},
                               {"s2 dW 128x512 @60x80", 512, 128, 8, 4800}, {"s1 dW 256x64 @120x160", 64, 256, 8, 19200}, {"fpn dW 256x256 @120x160", 256, 256, 8, 19200},
                               {"s4 dW 2048x512 @15x20", 512, 2048, 8, 304}, {"tail dW 200x72 hw 1216", 72, 200, 3, 1216}};
  hipStream_t st; CK(hipStreamCreate(&st));
  float* flushbuf = nullptr; if (cold) CK(hipMalloc(&flushbuf, 512ll << 20));
  printf("%-26s %7s | %8s %7s %5s | %8s %7s | %s\n", "shape", "GFLOP", "split us", "TF/s eq", "S", "fp32 us", "TF/s", "error / sum|a||b|: split max rms | fp32 max rms");
  for (const Shape& s : shapes) {
    if (filter && !strstr(s.name, filter)) continue;
    const long long nx = (long long)s.B * s.C * s.HW, ny = (long long)s.B * s.M * s.HW, nw = (long long)s.M * s.C;
    float *x = dalloc(nx), *dy = dalloc(ny), *dw = dalloc(nw), *dwo = dalloc(nw);
    fill_kernel<<<(unsigned)((nx + 255) / 256), 256>>>(x, nx, 1u, 1.f); fill_kernel<<<(unsigned)((ny + 255) / 256), 256>>>(dy, ny, 2u, 1e-3f);
    const int tilesM = (s.M + 127) / 128, tilesC = (s.C + 127) / 128, chunks = s.B * s.HW / 16;
    int splits = wgs / (tilesM * tilesC); if (splits < 1) splits = 1; if (splits > chunks / 4) splits = chunks / 4 > 0 ? chunks / 4 : 1;
    float* part = dalloc((long long)splits * nw);
    WgSplitArgs a; a.x = x; a.dy = dy; a.part = part; a.B = s.B; a.C = s.C; a.M = s.M; a.HW = s.HW; a.tilesM = tilesM; a.tilesC = tilesC; a.splits = splits; a.chunks = chunks;
    auto run_split = [&]() {
      if (dbg & 1) wgrad_split_kernel<1><<<dim3(tilesM * tilesC, splits), 256, 0, st>>>(a); else wgrad_split_kernel<0><<<dim3(tilesM * tilesC, splits), 256, 0, st>>>(a);
      wgrad_reduce_kernel<<<(unsigned)((nw + 255) / 256), 256, 0, st>>>(part, dw, nw, splits);
    };
    prn_conv_desc c; memset(&c, 0, sizeof(c));
    c.B = s.B; c.C = s.C; c.H = 1; c.W = s.HW; c.M = s.M; c.KH = c.KW = 1; c.stride = 1; c.pad = 0; c.Ho = 1; c.Wo = s.HW; c.in_mode = PRN_IN_ZERO; c.dil = 1; c.ystride = 1;
    const long long owsb = prn_conv2d_wgrad_ws_bytes(&c);
    float* ows = owsb > 0 ? dalloc(owsb / 4) : nullptr;
    auto run_old = [&]() { if (prn_conv2d_wgrad(&c, x, dy, dwo, ows, st)) { printf("old: %s\n", prn_last_error()); exit(1); } };
    auto timeit = [&](auto fn) {
      fn(); CK(hipStreamSynchronize(st));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      if (cold) {
        float tot = 0.f;
        for (int i = 0; i < 10; ++i) {
          flush_kernel<<<(512 << 20) / 4 / 256, 256, 0, st>>>(flushbuf, (512ll << 20) / 4, (float)i);
          CK(hipEventRecord(e0, st)); fn(); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
        }
        return tot / 10 * 1e3f;
      }
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) fn();
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      return ms / reps * 1e3f;
    };
    const float tsp = timeit(run_split), told = timeit(run_old);
    const double gf = 2.0 * s.M * s.C * (double)s.B * s.HW / 1e9;
    double *ref, *mag; CK(hipMalloc(&ref, nw * 8)); CK(hipMalloc(&mag, nw * 8));
    ref64_kernel<<<(unsigned)((nw + 255) / 256), 256, 0, st>>>(x, dy, ref, mag, s.B, s.C, s.M, s.HW);
    CK(hipStreamSynchronize(st));
    std::vector<float> h1(nw), h2(nw); std::vector<double> hr(nw), hm(nw);
    CK(hipMemcpy(h1.data(), dw, nw * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), dwo, nw * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), ref, nw * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hm.data(), mag, nw * 8, hipMemcpyDeviceToHost));
    double mx[2] = {0, 0}, sq[2] = {0, 0};
    for (long long i = 0; i < nw; ++i) { const double d = hm[i] + 1e-300, e0 = (h1[i] - hr[i]) / d, e1 = (h2[i] - hr[i]) / d; if (fabs(e0) > mx[0]) mx[0] = fabs(e0); if (fabs(e1) > mx[1]) mx[1] = fabs(e1); sq[0] += e0 * e0; sq[1] += e1 * e1; }
    printf("%-26s %7.2f | %8.1f %7.1f %5d | %8.1f %7.1f | %.2e %.2e | %.2e %.2e\n", s.name, gf, tsp, gf / tsp * 1e3, splits, told, gf / told * 1e3, mx[0], sqrt(sq[0] / nw), mx[1], sqrt(sq[1] / nw));
    fflush(stdout);
    hipFree(x); hipFree(dy); hipFree(dw); hipFree(dwo); hipFree(part); hipFree(ref); hipFree(mag); if (ows) hipFree(ows);
  }
  return 0;
}

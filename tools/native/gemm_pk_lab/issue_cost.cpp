// What does one instruction of kind X cost a wave whose matrix pipe is otherwise saturated?  One 256-thread workgroup per CU
// (one wave per SIMD), a loop of 8 independent fp32 MFMAs (512 cycles of pipe time) with N instructions of kind X placed between
// them; cycles per iteration from s_memtime.  Kinds: 0 nothing, 1 LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave),
// 2 buffer_load_dwordx4 to VGPRs (consumed later by a ds_write_b128), 3 ds_read_b32 x4, 4 ds_write_b128, 5 s_barrier,
// 6 buffer_load_dwordx4 to VGPRs + its ds_write_b128 one iteration later.
//   hipcc --offload-arch=gfx950 -O3 tools/native/issue_cost.cpp -o tools/native/issue_cost.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_dma16(unsigned lds_byte_addr, i32x4 desc, unsigned voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_byte_addr), "v"(voff), "s"(desc), "s"(soff) : "memory");
}

template <int KIND, int NX, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe(const float* src, float* out, unsigned long long* cyc, int iters, int bytes) {
  __shared__ __attribute__((aligned(16))) float smem[16384];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a = src[tid], b = src[tid + 64];
  const unsigned long long q = (unsigned long long)src;
  i32x4 d; d.x = (int)(unsigned)q; d.y = (int)(unsigned)(q >> 32) & 0xffff; d.z = bytes; d.w = 0x00020000;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem + w * 4096 * 4;
  unsigned voff = (blockIdx.x * 256 + tid) * 16;
  f32x4 hold[NX > 0 ? NX : 1];
  for (int i = 0; i < (NX > 0 ? NX : 1); ++i) hold[i] = f32x4{0, 0, 0, 0};
  float sink = 0.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma nounroll
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (m < NX) {
        if (KIND == 1) lds_dma16(lds0 + m * 1024, d, voff, (it & 63) * 65536);
        if (KIND == 2) hold[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (it & 63) * 65536 + m * 4096, 0));
        if (KIND == 3) { sink += smem[(w * 4096 + lane + m * 64 + it) & 16383]; sink += smem[(w * 4096 + lane + m * 64 + 1024 + it) & 16383]; }
        if (KIND == 4) *reinterpret_cast<f32x4*>(&smem[w * 4096 + lane * 4 + m * 256]) = f32x4{sink, a, b, sink};
        if (KIND == 5) __builtin_amdgcn_s_barrier();
        if (KIND == 6) {
          *reinterpret_cast<f32x4*>(&smem[w * 4096 + lane * 4 + m * 256]) = hold[m];      // last iteration's load
          hold[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (it & 63) * 65536 + m * 4096, 0));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = sink;
  for (int j = 0; j < 4; ++j) s += acc[j][0];
  if (KIND == 2) for (int i = 0; i < NX; ++i) s += hold[i].x;
  out[blockIdx.x * blockDim.x + tid] = s + smem[tid];
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int NX, int WAVES>
void run(const char* name, const float* src, float* out, unsigned long long* cyc, int bytes, int blocks) {
  const int iters = 2000;
  probe<KIND, NX, WAVES><<<blocks, 64 * WAVES>>>(src, out, cyc, iters, bytes);
  probe<KIND, NX, WAVES><<<blocks, 64 * WAVES>>>(src, out, cyc, iters, bytes);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(blocks);
  CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
  double m = 0;
  for (auto v : h) m += (double)v;
  m /= blocks * (double)iters;
  printf("%-44s waves/CU %d blocks %4d  x%d : %7.1f cycles / 8 MFMA  (+%6.1f, %5.1f per op)\n", name, WAVES, blocks, NX, m, m - 512.0, NX ? (m - 512.0) / NX : 0.0);
}

int main() {
  const int bytes = 256 << 20;
  float *src, *out; unsigned long long* cyc;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&out, 2048 * 512 * 4)); CK(hipMalloc(&cyc, 2048 * 8));
  CK(hipMemset(src, 0, bytes));
  run<0, 0, 4>("MFMA only", src, out, cyc, bytes, 256);
  run<0, 0, 8>("MFMA only", src, out, cyc, bytes, 256);
  run<1, 1, 4>("LDS-DMA 1 KiB", src, out, cyc, bytes, 256);
  run<1, 2, 4>("LDS-DMA 1 KiB", src, out, cyc, bytes, 256);
  run<1, 4, 4>("LDS-DMA 1 KiB", src, out, cyc, bytes, 256);
  run<1, 4, 8>("LDS-DMA 1 KiB", src, out, cyc, bytes, 256);
  run<1, 4, 4>("LDS-DMA 1 KiB (1 block only)", src, out, cyc, bytes, 1);
  run<2, 1, 4>("buffer_load_dwordx4 -> VGPR", src, out, cyc, bytes, 256);
  run<2, 2, 4>("buffer_load_dwordx4 -> VGPR", src, out, cyc, bytes, 256);
  run<2, 4, 4>("buffer_load_dwordx4 -> VGPR", src, out, cyc, bytes, 256);
  run<2, 4, 8>("buffer_load_dwordx4 -> VGPR", src, out, cyc, bytes, 256);
  run<2, 4, 4>("buffer_load_dwordx4 -> VGPR (1 block only)", src, out, cyc, bytes, 1);
  run<6, 4, 4>("buffer_load_dwordx4 + ds_write_b128 (pipelined)", src, out, cyc, bytes, 256);
  run<6, 4, 8>("buffer_load_dwordx4 + ds_write_b128 (pipelined)", src, out, cyc, bytes, 256);
  run<3, 4, 4>("2 x ds_read_b32", src, out, cyc, bytes, 256);
  run<3, 8, 4>("2 x ds_read_b32", src, out, cyc, bytes, 256);
  run<4, 4, 4>("ds_write_b128", src, out, cyc, bytes, 256);
  run<5, 1, 4>("s_barrier", src, out, cyc, bytes, 256);
  run<5, 2, 4>("s_barrier", src, out, cyc, bytes, 256);
  run<5, 1, 8>("s_barrier", src, out, cyc, bytes, 256);
  return 0;
}

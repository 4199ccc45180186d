// Probe: the shader clock the chip actually sustains under a given instruction mix, against the 2.4 GHz the MFMA peak of the roofline assumes.
// Every wave reads the shader-clock counter (s_memtime) and the constant 100-MHz counter (s_memrealtime) around ~400 us of work:
//   clock = d(s_memtime) / d(s_memrealtime) x 100 MHz.   Modes: MFMA only (bf16 32x32x16, independent accumulators), VALU only (v_fma_f32),
// the attention-like mix (16 MFMAs + 64 VALU per round), all on 256 CUs x 8 waves.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, int trips, unsigned long long* res) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + ((threadIdx.x + j) & 7)); b[j] = (short)(0x3f00 + ((threadIdx.x * 3 + j) & 7)); }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * ((threadIdx.x + i) & 63);
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < trips; ++t) {
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k & 3], 0, 0, 0);
    }
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int r = 0; r < (MODE == 1 ? 8 : 4); ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 128 && threadIdx.x == 0) { res[0] = c1 - c0; res[1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, int trips) {
  float* out; unsigned long long* res;
  CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&res, 16));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, out, trips, res);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CHECK(hipMemcpy(h, res, 16, hipMemcpyDeviceToHost));
    printf("%-34s run %d: %8.1f us by events, %9llu shader cycles in %7.1f us of the 100-MHz counter -> %.3f GHz\n", name, rep, ms * 1e3, h[0], h[1] / 100.0,
           (double)h[0] / (h[1] / 100.0) * 1e-3);
  }
  CHECK(hipFree(out)); CHECK(hipFree(res));
}
int main() {
  run<1>("VALU only (v_fma_f32)", 6000);
  run<0>("MFMA only (bf16 32x32x16)", 1500);
  run<2>("16 MFMAs + 64 VALU per round", 1000);
  run<1>("VALU only again", 6000);
  return 0;
}

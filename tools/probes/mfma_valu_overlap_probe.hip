// Feasibility probe (not product code): can ONE wave overlap its own MFMAs with independent VALU / transcendental work
// when the instruction stream interleaves them (sched_group_barrier), and how much does it buy over "all MFMAs, then all
// VALU"?  Decides whether the attention kernel should software-pipeline softmax(t) under the QK^T MFMAs of tile t+1.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap_probe mfma_valu_overlap_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// per loop trip: NM MFMAs (32x32x16 bf16, 4 independent accumulators) and NV VALU ops per lane, of which every 4th is v_exp
template <int MODE, int NM, int NV>
__global__ __launch_bounds__(256) void probe(float* out, int trips, unsigned long long* cyc) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x + i)); b[i] = (__bf16)(0.02f * i); }
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < trips; ++t) {
    if (MODE != 2) {
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
    }
    if (MODE != 1) {
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int i = k & 15;
        if ((k & 3) == 3) v[i] = __builtin_amdgcn_exp2f(v[i] * 0.5f);
        else v[i] = fmaf(v[i], 0.999f, 0.001f);
      }
    }
    if (MODE == 3) {        // interleave: 1 MFMA then NV/NM VALU, repeated
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV / NM, 0);
      }
    }
    if (MODE == 0) __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int NM, int NV>
void run(const char* name, int blocks_per_cu) {
  float* out; unsigned long long* cyc;
  const int blocks = 256 * blocks_per_cu, trips = 2000;
  CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  CHECK(hipMalloc(&cyc, 8));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe<MODE, NM, NV>), dim3(blocks), dim3(256), 0, 0, out, trips, cyc);
    CHECK(hipDeviceSynchronize());
  }
  unsigned long long h;
  CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
  printf("%-44s NM=%2d NV=%3d waves/SIMD=%d : %7.1f cycles per trip (MFMA alone would be %d)\n", name, NM, NV, blocks_per_cu, (double)h / trips, NM * 32);
  CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<1, 16, 128>("MFMA only", w);
    run<2, 16, 128>("VALU only", w);
    run<0, 16, 128>("MFMA block, then VALU block (fenced)", w);
    run<3, 16, 128>("interleaved 1 MFMA : 8 VALU (sched_group)", w);
    run<3, 16, 64>("interleaved 1 MFMA : 4 VALU (sched_group)", w);
    run<0, 16, 64>("MFMA block, then VALU block (fenced)", w);
  }
  return 0;
}

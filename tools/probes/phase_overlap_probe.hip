// Probe (not product code): 8 waves per workgroup, 2 per SIMD.  Per phase, waves 0-3 run a matrix segment (16 MFMAs
// 32x32x16 bf16) while waves 4-7 run a VALU segment, then the roles swap; one s_barrier per phase -- the skeleton of
// attention8_bf16_kernel.  Which VALU mixes overlap with the SIMD-mate's MFMAs, and which serialise?
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/phase_overlap_probe.hip -o /tmp/phase_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MIX: 0 = 128 fma; 1 = 32 fma + 32 exp + 32 add + 16 cvt_pk (softmax block); 2 = 32 exp only; 3 = 96 fma (no exp, same count as 1
// without the exps + cvt); 4 = 32 fma + 32 add + 16 cvt (softmax without exp); 5 = nothing (MFMA side alone)
template <int MIX>
__device__ __forceinline__ void valu_seg(float (&v)[32], unsigned (&pk)[16]) {
  if (MIX == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 32; ++i) v[i] = fmaf(v[i], 0.999f, 0.001f);
  } else if (MIX == 1 || MIX == 4) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      float x = fmaf(v[i], 0.18f, -0.3f);
      if (MIX == 1) x = __builtin_amdgcn_exp2f(x);
      s += x;
      v[i] = x;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
      typedef __attribute__((ext_vector_type(2))) float f2;
      f2 p = {v[2 * i], v[2 * i + 1]};
      pk[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(p, bf2));
    }
    v[0] += s * 1e-9f;
  } else if (MIX == 2) {
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
  } else if (MIX == 3) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < 32; ++i) v[i] = fmaf(v[i], 0.999f, 0.001f);
  }
}

template <int MIX, bool MFMA>
__global__ __launch_bounds__(512) void probe(float* out, int trips, unsigned long long* cyc) {
  bf16x8 a[4], b[4];
  for (int q = 0; q < 4; ++q) for (int i = 0; i < 8; ++i) { a[q][i] = (__bf16)(0.01f * ((threadIdx.x + i + q) & 31)); b[q][i] = (__bf16)(0.02f * (i + q)); }
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float v[32]; unsigned pk[16];
  for (int i = 0; i < 32; ++i) v[i] = 0.001f * ((threadIdx.x + i) & 63);
  for (int i = 0; i < 16; ++i) pk[i] = 0;
  const bool grpB = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) != 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      if (grpB == (ph == 1)) {
        if (MFMA) {
#pragma unroll
          for (int m = 0; m < 16; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 3], b[(m >> 2) & 3], acc[m & 3], 0, 0, 0);
        }
      } else {
        valu_seg<MIX>(v, pk);
      }
      __builtin_amdgcn_s_barrier();
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  for (int i = 0; i < 32; ++i) s += v[i];
  for (int i = 0; i < 16; ++i) s += (float)pk[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// Same work, but EVERY wave runs one stream that interleaves its own 16 MFMAs with its own VALU segment (the work of one
// "slot" of a software-pipelined loop); ILV = VALU-class instructions the scheduler is asked to place after each MFMA
// (0 = MFMA block then VALU block, fenced).  One barrier per slot.
template <int MIX, int ILV>
__global__ __launch_bounds__(512) void probe_self(float* out, int trips, unsigned long long* cyc) {
  bf16x8 a[4], b[4];
  for (int q = 0; q < 4; ++q) for (int i = 0; i < 8; ++i) { a[q][i] = (__bf16)(0.01f * ((threadIdx.x + i + q) & 31)); b[q][i] = (__bf16)(0.02f * (i + q)); }
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float v[32]; unsigned pk[16];
  for (int i = 0; i < 32; ++i) v[i] = 0.001f * ((threadIdx.x + i) & 63);
  for (int i = 0; i < 16; ++i) pk[i] = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 3], b[(m >> 2) & 3], acc[m & 3], 0, 0, 0);
    if (ILV == 0) __builtin_amdgcn_sched_barrier(0);
    valu_seg<MIX>(v, pk);
    if (ILV > 0) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x402, ILV, 0);     // VALU + transcendental
      }
    }
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  for (int i = 0; i < 32; ++i) s += v[i];
  for (int i = 0; i < 16; ++i) s += (float)pk[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MIX, int ILV>
void run_self(const char* name, int threads) {
  float* out; unsigned long long* cyc;
  const int blocks = 256, trips = 2000;
  CHECK(hipMalloc(&out, (size_t)blocks * 512 * 4));
  CHECK(hipMalloc(&cyc, 8));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe_self<MIX, ILV>), dim3(blocks), dim3(threads), 0, 0, out, trips, cyc);
    CHECK(hipDeviceSynchronize());
  }
  unsigned long long h;
  CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
  printf("%-58s %d waves/SIMD: %7.1f cycles per slot = %6.1f per SIMD per (16 MFMA + segment)\n", name, threads / 256, (double)h / trips, (double)h / trips / (threads / 256));
  CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

template <int MIX, bool MFMA>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  const int blocks = 256, trips = 2000;
  CHECK(hipMalloc(&out, (size_t)blocks * 512 * 4));
  CHECK(hipMalloc(&cyc, 8));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe<MIX, MFMA>), dim3(blocks), dim3(512), 0, 0, out, trips, cyc);
    CHECK(hipDeviceSynchronize());
  }
  unsigned long long h;
  CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
  printf("%-64s : %7.1f cycles per phase (16 MFMAs alone = 512)\n", name, (double)h / trips / 2);
  CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main() {
  run<5, true>("MFMA segment || nothing");
  run<0, false>("nothing || 128 fma");
  run<0, true>("MFMA segment || 128 fma");
  run<3, false>("nothing || 96 fma");
  run<3, true>("MFMA segment || 96 fma");
  run<2, false>("nothing || 32 exp");
  run<2, true>("MFMA segment || 32 exp");
  run<4, false>("nothing || 32 fma + 32 add + 16 cvt_pk");
  run<4, true>("MFMA segment || 32 fma + 32 add + 16 cvt_pk");
  run<1, false>("nothing || softmax block (32 fma, 32 exp, 32 add, 16 cvt_pk)");
  run<1, true>("MFMA segment || softmax block (32 fma, 32 exp, 32 add, 16 cvt_pk)");
  for (int th = 256; th <= 512; th += 256) {
    run_self<1, 0>("own stream: 16 MFMA, then softmax block (fenced)", th);
    run_self<1, 4>("own stream: softmax block interleaved 4 per MFMA", th);
    run_self<1, 7>("own stream: softmax block interleaved 7 per MFMA", th);
    run_self<4, 5>("own stream: 32 fma + 32 add + 16 cvt interleaved 5 per MFMA", th);
    run_self<3, 6>("own stream: 96 fma interleaved 6 per MFMA", th);
  }
  return 0;
}

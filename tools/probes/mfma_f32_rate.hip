// Probe: issue rate of v_mfma_f32_32x32x2_f32 per SIMD for dependent / independent accumulator chains, operands from registers or LDS,
// one or two waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f32_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int NACC, bool LDS>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  __shared__ float buf[64 * 33];
  for (int i = threadIdx.x; i < 64 * 33; i += blockDim.x) buf[i] = 1.0f + i * 1e-6f;
  __syncthreads();
  const int lane = threadIdx.x & 63, ln = lane & 31, hf = lane >> 5;
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int j = 0; j < 16; ++j) acc[a][j] = 0.f;
  float b[32];
  for (int i = 0; i < 32; ++i) b[i] = 1.f + 0.001f * (i + lane);
  for (int it = 0; it < iters; ++it) {
    float a[32];
    if (LDS) {
#pragma unroll
      for (int i = 0; i < 32; ++i) a[i] = buf[(2 * i + hf) * 33 + ((ln + it) & 31)];
    } else {
#pragma unroll
      for (int i = 0; i < 32; ++i) a[i] = b[(i + 1) & 31];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i % NACC] = MF(a[i], b[i], acc[i % NACC]);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int j = 0; j < 16; ++j) s += acc[a][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(const char* name, int threads, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, LDS>), dim3(256), dim3(threads), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, LDS>), dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 32 * (threads / 64) / 4;
  const double ns = ms * 1e6 / mfma_per_simd;
  printf("%-44s %d waves/SIMD: %.1f ns per MFMA and SIMD = %.0f cycles at 2.4 GHz, %.1f TFLOP/s\n", name, threads / 256, ns, ns * 2.4,
         256.0 * 4 * 4096 / ns / 1e3);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  run<1, false>("1 accumulator, operands in registers", 256, out);
  run<1, false>("1 accumulator, operands in registers", 512, out);
  run<2, false>("2 accumulators, operands in registers", 256, out);
  run<2, false>("2 accumulators, operands in registers", 512, out);
  run<4, false>("4 accumulators, operands in registers", 256, out);
  run<1, true>("1 accumulator, A from LDS (32 ahead)", 256, out);
  run<1, true>("1 accumulator, A from LDS (32 ahead)", 512, out);
  run<2, true>("2 accumulators, A from LDS (32 ahead)", 512, out);
  return 0;
}

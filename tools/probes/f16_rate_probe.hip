// Issue rates on gfx950: v_cvt_pk_f16_f32 vs v_cvt_pk_bf16_f32, v_mfma_f32_32x32x16_f16 vs _bf16 (one wave per SIMD, dependent-free streams).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/f16_rate_probe.hip -o /tmp/f16_rate_probe && /tmp/f16_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters) {
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
  unsigned acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  s8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { f32x2 f = {x[2 * i], x[2 * i + 1]}; acc[i] ^= __builtin_bit_cast(unsigned, __builtin_convertvector(f, h2)); asm volatile("" : "+v"(x[2 * i])); }
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { f32x2 f = {x[2 * i], x[2 * i + 1]}; acc[i] ^= __builtin_bit_cast(unsigned, __builtin_convertvector(f, b2)); asm volatile("" : "+v"(x[2 * i])); }
    } else if (MODE == 2) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c3, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / iters;
}
int main() {
  float* d; hipMalloc(&d, 1 << 20);
  const char* names[4] = {"8 x v_cvt_pk_f16_f32", "8 x v_cvt_pk_bf16_f32", "4 x v_mfma_f32_32x32x16_f16", "4 x v_mfma_f32_32x32x16_bf16"};
  for (int waves = 1; waves <= 2; ++waves)
    for (int m = 0; m < 4; ++m) {
      dim3 g(256), bl(256 * waves);
      if (m == 0) hipLaunchKernelGGL(k<0>, g, bl, 0, 0, d, 4096);
      if (m == 1) hipLaunchKernelGGL(k<1>, g, bl, 0, 0, d, 4096);
      if (m == 2) hipLaunchKernelGGL(k<2>, g, bl, 0, 0, d, 4096);
      if (m == 3) hipLaunchKernelGGL(k<3>, g, bl, 0, 0, d, 4096);
      float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
      printf("%d wave(s)/SIMD  %-30s %.1f cycles per iteration\n", waves, names[m], h);
    }
  return 0;
}

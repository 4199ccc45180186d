// Probe: issue cost (cycles per wave64 instruction, one wave per SIMD and two) of the VALU instructions the attention
// softmax is made of.  build: hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate_probe.hip -o /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void probe(float* out, int trips, unsigned long long* cyc) {
  float v[16]; f2 w[8];
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * ((threadIdx.x + i) & 63);
  for (int i = 0; i < 8; ++i) w[i] = f2{v[2 * i], v[2 * i + 1]};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        if (OP == 1) asm volatile("v_exp_legacy_f32 %0, %0" : "+v"(v[i]));
        if (OP == 2) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
        if (OP == 3) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[i]));
        if (OP == 4) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
        if (OP == 5) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(w[i & 7]));
        if (OP == 6) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(w[i & 7]));
        if (OP == 7) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v[i]));
        if (OP == 8) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(v[i]));
        if (OP == 9) asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(v[i]));
        if (OP == 10) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
        if (OP == 11) asm volatile("v_ldexp_f32 %0, %0, %0" : "+v"(v[i]));
      }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += w[i].x + w[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
void run(const char* name) {
  for (int threads = 256; threads <= 512; threads += 256) {
    float* out; unsigned long long* cyc;
    const int trips = 500;
    CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&cyc, 8));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(threads), 0, 0, out, trips, cyc); CHECK(hipDeviceSynchronize()); }
    unsigned long long h; CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-20s %d wave(s)/SIMD: %6.2f cycles per instruction per wave, %6.2f per SIMD-issued instruction\n", name, threads / 256, (double)h / trips / 64, (double)h / trips / 64 / (threads / 256));
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
  }
}
int main() {
  run<0>("v_exp_f32"); run<1>("v_exp_legacy_f32"); run<2>("v_exp_f16"); run<3>("v_add_f32"); run<4>("v_fma_f32"); run<5>("v_pk_add_f32");
  run<6>("v_pk_fma_f32"); run<7>("v_cvt_pk_bf16_f32"); run<8>("v_max3_f32"); run<9>("v_mul_lo_u32"); run<10>("v_rcp_f32"); run<11>("v_ldexp_f32");
  return 0;
}

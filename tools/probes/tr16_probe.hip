// Probe: semantics of ds_read_b64_tr_b16 with arbitrary per-lane addresses (gfx950).
// Hypothesis: within each 16-lane group, lane i receives, as element j (0..3), element (i & 3) of the 8-byte chunk
// addressed by lane 4*j + (i >> 2) of the same group.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr16_probe.hip -o gpurun_out/tr16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const int* chunk_of_lane, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + 4 * chunk_of_lane[threadIdx.x]));
  *(v4s*)(out + threadIdx.x * 4) = r;
}
int main() {
  int h[64]; short o[256];
  srand(7);
  for (int trial = 0; trial < 3; ++trial) {
    for (int l = 0; l < 64; ++l) h[l] = trial == 0 ? l : (trial == 1 ? (l * 37 + 11) % 2048 : rand() % 2048);
    int* d; short* od;
    hipMalloc(&d, sizeof(h)); hipMalloc(&od, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, od);
    hipMemcpy(o, od, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const int g = l >> 4, i = l & 15;
        const int src_lane = 16 * g + 4 * j + (i >> 2);
        const int expect = 4 * h[src_lane] + (i & 3);
        if (o[4 * l + j] != (short)expect) { if (bad < 8) printf("trial %d lane %d elem %d: got %d expected %d\n", trial, l, j, o[4 * l + j], expect); ++bad; }
      }
    printf("trial %d: %s (%d mismatches)\n", trial, bad ? "HYPOTHESIS WRONG" : "hypothesis holds", bad);
  }
  return 0;
}

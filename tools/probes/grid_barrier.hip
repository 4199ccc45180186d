// Probe: cost of a grid-wide barrier (release / acquire at agent scope through one counter) for W co-resident workgroups of 512 threads, with a
// 2 MB exchange written before and read after every barrier (the pattern of one Sinkhorn iteration of the training step).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/grid_barrier.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

__global__ __launch_bounds__(512) void k(unsigned* counter, float* buf, int cols, int iters, int exchange) {
  const int W = gridDim.x;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (exchange)
      for (int j = threadIdx.x; j < cols; j += 512) buf[(size_t)blockIdx.x * cols + j] = acc + j;
    grid_barrier(counter, (unsigned)(2 * it + 1) * W);
    if (exchange) {
      // each workgroup folds a stripe of columns over all W partials
      const int c = blockIdx.x * 8 + (threadIdx.x & 7), g = threadIdx.x >> 3;
      float s = 0.f;
      if (c < cols)
        for (int b = g; b < W; b += 64) s += __builtin_nontemporal_load(buf + (size_t)b * cols + c);
      acc += s;
    }
    grid_barrier(counter, (unsigned)(2 * it + 2) * W);
  }
  if (acc == 12345.f) buf[0] = acc;
}

int main() {
  unsigned* counter; float* buf;
  hipMalloc(&counter, 4); hipMalloc(&buf, (size_t)512 * 4096 * 4);
  for (int W : {128, 257, 512}) {
    for (int exchange = 0; exchange < 2; ++exchange) {
      const int iters = 200;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      void* args[] = {&counter, &buf, (void*)nullptr, (void*)nullptr, (void*)nullptr};
      int cols = 2049, it = iters, ex = exchange;
      args[2] = &cols; args[3] = &it; args[4] = &ex;
      hipMemset(counter, 0, 4);
      hipError_t rc = hipLaunchCooperativeKernel((const void*)k, dim3(W), dim3(512), args, 0, 0);
      hipDeviceSynchronize();
      hipMemset(counter, 0, 4);
      hipEventRecord(e0);
      rc = hipLaunchCooperativeKernel((const void*)k, dim3(W), dim3(512), args, 0, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("W=%d exchange=%d: rc=%d  %.2f us per iteration (two barriers%s)\n", W, exchange, (int)rc, ms * 1e3 / iters, exchange ? " + 2 MB write / fold" : "");
    }
  }
  return 0;
}

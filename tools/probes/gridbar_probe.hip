// Feasibility probe (not product code): cost of the grid-wide exchange step a persistent (on-chip) Sinkhorn would need
// on MI355X.  256 co-resident workgroups; per iteration: publish column partials (write-through 16-byte stores),
// barrier, distributed reduction of the partials, publish, barrier, broadcast read.
//   config A: 2 problems x 128 slabs x 4096 columns, chip-wide barrier (flat or hierarchical)
//   config B: per XCD 4 problems x 8 slabs x 1024 columns, barrier among the 32 workgroups of one XCD only
// build: hipcc --offload-arch=gfx950 -O3 -o gridbar_probe gridbar_probe.hip ; run: ./gridbar_probe [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_agent(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4 ld4_agent(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// counters: 64 unsigned apart (own 256-byte line each): [0] global, [64*(1+x)] per XCD
template <int KIND>   // 0 flat chip-wide, 1 hierarchical chip-wide, 2 XCD-local
__device__ __forceinline__ void grid_barrier(unsigned* cnt, unsigned epoch, int nb, int* fail) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int x = blockIdx.x & 7, per = nb / 8;
    unsigned* wait_on = cnt;
    unsigned target = epoch * (KIND == 0 ? nb : 8);
    if (KIND == 0) {
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (KIND == 1) {
      const unsigned old = __hip_atomic_fetch_add(cnt + 64 * (1 + x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == epoch * per - 1) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      wait_on = cnt + 64 * (1 + x);
      target = epoch * per;
      __hip_atomic_fetch_add(wait_on, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int spins = 0;
    while (__hip_atomic_load(wait_on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { *fail = 1; break; }     // bounded: never hang the GPU
    }
  }
  __syncthreads();
}

// NSLAB slabs (workgroups) per problem, NCOL columns; thread t owns NCOL/512 columns as float4 groups when >= 4
template <int KIND, int NSLAB, int NCOL, bool EXCH>
__global__ __launch_bounds__(512) void probe(float* partials, float* v, unsigned* cnt, int* fail, int iters, float* out) {
  const int b = blockIdx.x, t = threadIdx.x, nb = gridDim.x;
  // blocks of one problem: KIND 2 -> same XCD (b & 7), consecutive local ids; else consecutive b
  const int local = KIND == 2 ? (b >> 3) : b;
  const int prob = (KIND == 2 ? (b & 7) * (nb / 8 / NSLAB) : 0) + local / NSLAB, slab = local % NSLAB;
  constexpr int CPT = NCOL / 512;              // columns per thread (2 or 8)
  constexpr int RC = NCOL / NSLAB;             // columns reduced by one block
  float acc = 0.f;
  unsigned epoch = 0;
  float vv[CPT];
  for (int c = 0; c < CPT; ++c) vv[c] = 1.f;
  __shared__ float red[512];
  for (int it = 0; it < iters; ++it) {
    if (EXCH) {
      float* mine = partials + ((size_t)(prob * NSLAB + slab)) * NCOL + t * CPT;
      if (CPT >= 4) for (int c = 0; c < CPT; c += 4) st4_agent(mine + c, f32x4{vv[c] + it, vv[c + 1], vv[c + 2], vv[c + 3]});
      else for (int c = 0; c < CPT; ++c) st_agent(mine + c, vv[c] + it);
    }
    grid_barrier<KIND>(cnt, ++epoch, nb, fail);
    if (EXCH) {
      // reduce RC columns over NSLAB slabs with 512 threads: thread -> (column group of 4, slab subset)
      constexpr int G4 = RC / 4;               // float4 groups of columns
      constexpr int SS = 512 / G4;             // slab subsets
      const int g4 = t % G4, ss = t / G4;
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      if (ss < SS)
        for (int k = ss; k < NSLAB; k += SS) s += ld4_agent(partials + ((size_t)(prob * NSLAB + k)) * NCOL + slab * RC + 4 * g4);
      // cross-subset reduction through LDS (only first component lanes matter for timing realism)
      red[t] = s.x + s.y + s.z + s.w;
      __syncthreads();
      if (t < G4) {
        float tot = 0.f;
        for (int k = 0; k < SS; ++k) tot += red[t + G4 * k];
        st4_agent(v + prob * NCOL + slab * RC + 4 * t, f32x4{tot * 1e-3f, 1.f, 1.f, 1.f});
      }
    }
    grid_barrier<KIND>(cnt, ++epoch, nb, fail);
    if (EXCH) {
      if (CPT >= 4) for (int c = 0; c < CPT; c += 4) { f32x4 r = ld4_agent(v + prob * NCOL + t * CPT + c); vv[c] = r.x; vv[c + 1] = r.y; vv[c + 2] = r.z; vv[c + 3] = r.w; }
      else for (int c = 0; c < CPT; ++c) vv[c] = ld_agent(v + prob * NCOL + t * CPT + c);
    }
    acc += vv[0];
  }
  if (t == 0) out[b] = acc;
}

template <typename K>
void run(const char* name, K kern, float* partials, float* v, unsigned* cnt, int* fail, int iters, float* out) {
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipMemset(cnt, 0, 64 * 9 * 4));
    CHECK(hipMemset(fail, 0, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, partials, v, cnt, fail, iters, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    int f;
    CHECK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
    float o[256];
    CHECK(hipMemcpy(o, out, 256 * 4, hipMemcpyDeviceToHost));
    printf("%-58s %.2f us/iteration (2 barriers each), fail=%d, out[0]=%g out[255]=%g\n", name, ms * 1e3 / iters, f, o[0], o[255]);
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  float *partials, *v, *out;
  unsigned* cnt;
  int* fail;
  CHECK(hipMalloc(&partials, (size_t)256 * 4096 * 4));
  CHECK(hipMalloc(&v, 32 * 4096 * 4));
  CHECK(hipMalloc(&out, 256 * 4));
  CHECK(hipMalloc(&cnt, 64 * 9 * 4));
  CHECK(hipMalloc(&fail, 4));
  run("flat chip barrier only", probe<0, 128, 4096, false>, partials, v, cnt, fail, iters, out);
  run("hierarchical chip barrier only", probe<1, 128, 4096, false>, partials, v, cnt, fail, iters, out);
  run("XCD-local barrier only", probe<2, 8, 1024, false>, partials, v, cnt, fail, iters, out);
  run("A: flat chip barrier + 2x128x4096 exchange", probe<0, 128, 4096, true>, partials, v, cnt, fail, iters, out);
  run("A: hierarchical chip barrier + 2x128x4096 exchange", probe<1, 128, 4096, true>, partials, v, cnt, fail, iters, out);
  run("B: XCD-local barrier + (4/XCD)x8x1024 exchange", probe<2, 8, 1024, true>, partials, v, cnt, fail, iters, out);
  return 0;
}

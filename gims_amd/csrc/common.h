// Shared helpers for the gfx950 kernels of libgims_hip.so.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gims_hip.h"

namespace gims {

void set_error(const char* fmt, ...);

#define GIMS_CHECK_ARG(cond, ...)                    \
  do {                                               \
    if (!(cond)) {                                   \
      ::gims::set_error(__VA_ARGS__);                \
      return GIMS_EINVAL;                            \
    }                                                \
  } while (0)

#define GIMS_HIP(call)                                                                     \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      ::gims::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return GIMS_EHIP;                                                                    \
    }                                                                                      \
  } while (0)

#define GIMS_LAUNCH_CHECK()                                                                \
  do {                                                                                     \
    hipError_t e_ = hipGetLastError();                                                     \
    if (e_ != hipSuccess) {                                                                \
      ::gims::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
      return GIMS_EHIP;                                                                    \
    }                                                                                      \
  } while (0)

#define GIMS_LDS_ATTR(...) /* (function, bytes); variadic because template argument lists carry commas */ \
  do {                                                                                                     \
    const int rc_ = ::gims::lds_attr(__VA_ARGS__);                                                         \
    if (rc_ != GIMS_OK) return rc_;                                                                        \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// round-to-nearest-even f32 -> bf16 (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  f32x2 f = {lo, hi};
  hwbf16x2 h = __builtin_convertvector(f, hwbf16x2);
  return __builtin_bit_cast(uint32_t, h);
}
// round-to-nearest-even f32 -> IEEE half, saturated to the finite range (v_med3_f32 + v_cvt_pk_f16_f32)
typedef _Float16 hwf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_h2_sat(float lo, float hi) {
  f32x2 f = {__builtin_amdgcn_fmed3f(lo, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(hi, -65504.f, 65504.f)};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, hwf16x2));
}
__device__ __forceinline__ uint16_t f2bf(float x) { return (uint16_t)(pack_bf2(x, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// SPL32: the split-bf16 activation / weight layout.  A logical [rows][K] f32 matrix is stored as ONE bf16 buffer
// [rows][2K]: per row and per block of 32 channels, 32 hi values then 32 lo values (hi = bf16(x), lo = bf16(x - hi)),
// i.e. each (row, 32-channel block) is one full 128-byte line holding everything an MFMA k-step pair needs.
// Column k -> hi at spl_col(k), lo at spl_col(k) + 32.
__device__ __forceinline__ int spl_col(int k) { return ((k >> 5) << 6) + (k & 31); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// order-preserving map f32 -> u32 (ascending), for exact radix selection
__device__ __forceinline__ uint32_t f32_key(float x) {
  uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// Sinkhorn status word (0 ok, 1 numeric guard, 2 a bounded wait ran out): raised monotonically, device scope -- workgroups on
// different XCDs may report different codes for one problem and the larger one must survive whatever the write-back order
// (positive floats order like their bit patterns)
__device__ __forceinline__ void ot_raise_status(float* status, float code) {
  __hip_atomic_fetch_max((int*)status, __float_as_int(code), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// gims_attn_guard (include/gims_hip.h): does the statistic of the guarded launch ask for the redo?  Evaluated by every WAVE of a guarded launch at
// entry (the accumulator is final: the launch that filled it precedes this one on the stream); float64 arithmetic as the header states.  ONE
// load per lane -- lane i takes word i of the [n_heads + 1][4] table (n_heads <= 15) -- and shuffles: an unfired launch costs every wave a single
// L2 round trip (a per-head loop of dependent loads cost four).
__device__ __forceinline__ bool attn_guard_fires(const gims_attn_guard& g) {
  const unsigned long long* st = (const unsigned long long*)g.stat;
  const int lane = threadIdx.x & 63, nw = 4 * (g.n_heads + 1);
  const unsigned long long v = lane < nw ? __builtin_nontemporal_load(st + lane) : 0ull;
  bool mine = false;
  if (g.kind == GIMS_GUARD_PEAKED) {
    const unsigned long long cnt = __shfl(v, (lane & ~3) + 1, 64), tail = __shfl(v, (lane & ~3) + 3, 64);      // (lane 4h: v = the head's sum)
    const unsigned long long mx = __shfl(v, (lane & ~3) + 2, 64);
    if ((lane & 3) == 0 && lane < 4 * g.n_heads) {
      if (cnt != 0) {
        const double mean = (double)v / (double)cnt / 16777216.0, tl = (double)tail / (double)cnt;
        mine = mean > g.mean_thr || tl > g.tail_thr;
      }
      mine = mine || (g.max_thr > 0.0 && (double)mx / 16777216.0 >= g.max_thr);
    }
  } else if (lane >= 4 * g.n_heads && lane < 4 * g.n_heads + 3) {
    mine = !((double)__uint_as_float((uint32_t)v) <= g.range_limit);
  }
  return __ballot(mine) != 0ull;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// GIMS_GUARD_WALK=0: guarded launches take the full grid instead of one dispatch round of workgroups that walk the tiles (read per call: the
// cross-check of tests/test_hip_kernels.py::test_guarded_launches_that_walk_their_tiles switches it)
static inline bool guard_walk_enabled() { const char* e = getenv("GIMS_GUARD_WALK"); return !(e && atoi(e) == 0); }

// Host descriptor table -> device memory through KERNEL ARGUMENTS (chunks of <= 3968 bytes per launch): asynchronous on
// the stream, no staging buffer whose lifetime would need a synchronisation, no pageable-memory pinning by the runtime.
int upload_table(const void* host, size_t bytes, void* dev, hipStream_t stream);
// Process-wide state is keyed by DEVICE and created under a lock (no per-process function statics that would bind to whichever
// device was current at the first call).  lds_attr: hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (function, device).
// device_once: one device buffer per (key, device), optionally filled from `init` (synchronous copy, first call only); nullptr on failure.
// pinned_once: one pinned host buffer (zero-filled) per (key, device).
int lds_attr(const void* fn, int bytes);
void* device_once(const char* key, size_t bytes, const void* init);
void* pinned_once(const char* key, size_t bytes);
int current_device();
int device_cus();        // compute units of the current device (>= 8)
// linear6.hip: exact-class bf16x6 GEMM on SPL3 operands (batched), and its operand split
int linear_x6_batch_launch(const gims_linear_args* dev_args, int count, int max_m, int max_n, hipStream_t s);
int split_spl3_launch(const float* src, int64_t lds, uint16_t* dst, int64_t ldd, int64_t rows, int k, hipStream_t s);

// sinkhorn2d.hip: the 2-D on-chip Sinkhorn (all iterations in one launch; see that file's header)
struct OtR2Host { const float* z; int64_t ld; int n, m; float* u; float* v; float* status; float norm, log_mu_bin, log_nu_bin; };
struct OtR2Plan { bool ok; int nx, nc, ppg, ngroups; size_t bytes; };
OtR2Plan ot_res2_plan(const OtR2Host* pr, int np, int iters);
int ot_res2_run(const OtR2Plan& P, const OtR2Host* hp, int np, float alpha, int iters, int init_inside, char* base, hipStream_t s);

}  // namespace gims

// Fused multi-tensor Adam: the optimizer step of the reference's training loop (train.py:52-57 builds torch.optim.Adam over three
// parameter groups, :138 calls optimizer.step()) for ALL parameter tensors of the model in a handful of launches.
// HBM-bound elementwise work: per element 4 reads (p, g, m, v) + 3 writes (p, m, v) of 4 bytes; nothing to tile, nothing for the
// matrix cores -- the point is one launch per ~90 tensors instead of torch's ~90 launches + list marshalling per step (3 ms of host
// time per 29-ms step at 282 tensors).  Arithmetic follows torch's single-tensor Adam (torch/optim/adam.py, _single_tensor_adam,
// non-amsgrad, maximize = False, float32 op-math), operation by operation:
//     g' = g + wd * p                         (weight_decay != 0)
//     m  = m + (g' - m) * (1 - beta1)         (Tensor.lerp_, weight < 0.5 branch)
//     v  = v * beta2 + (1 - beta2) * g' * g'  (mul_ then addcmul_)
//     p  = p - step_size * m / (sqrt(v) / sqrt(1 - beta2^t) + eps),   step_size = lr / (1 - beta1^t)
// with the step-dependent scalars formed on the host in double and rounded to float once, as torch's scalar arguments are.
#include "common.h"

namespace gims {

constexpr int ADAM_MAX_TENSORS = 80;     // per launch: 80 * 36 + 81 * 4 + 8 * 28 bytes of kernel arguments (< 4 KB)
constexpr int ADAM_MAX_GROUPS = 8;
constexpr int ADAM_CHUNK = 4096;         // elements per workgroup (256 threads x 4 float4)

struct AdamGroup { float step_size, bc2_sqrt, beta2, eps, wd, one_m_beta1, one_m_beta2; };   // 1 - beta formed in double, like torch's scalars
struct AdamLaunch {
  float* p[ADAM_MAX_TENSORS];
  const float* g[ADAM_MAX_TENSORS];
  float* m[ADAM_MAX_TENSORS];
  float* v[ADAM_MAX_TENSORS];
  int first_chunk[ADAM_MAX_TENSORS + 1];   // prefix sum of chunk counts
  int n[ADAM_MAX_TENSORS];
  unsigned char group[ADAM_MAX_TENSORS];
  AdamGroup grp[ADAM_MAX_GROUPS];
  int count;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamGroup& h) {
  // no contraction across torch's separately rounded operations
  if (h.wd != 0.f) g = __fadd_rn(g, __fmul_rn(h.wd, p));
  m = __fadd_rn(m, __fmul_rn(__fsub_rn(g, m), h.one_m_beta1));
  v = __fmul_rn(v, h.beta2);
  v = __fadd_rn(v, __fmul_rn(__fmul_rn(h.one_m_beta2, g), g));      // addcmul_: input + value * t1 * t2
  const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), h.bc2_sqrt), h.eps);
  p = __fadd_rn(p, __fmul_rn(-h.step_size, __fdiv_rn(m, denom)));   // addcdiv_: input + value * (t1 / t2)
}

__global__ __launch_bounds__(256) void adam_kernel(const AdamLaunch a) {
  __shared__ int s_t;
  if (threadIdx.x < 64) {                       // which tensor owns this chunk: ballot over the prefix table
    const int b = (int)blockIdx.x;
    int t = -1;
    for (int base = 0; base < a.count && t < 0; base += 64) {
      const int i = base + (int)threadIdx.x;
      const bool mine = i < a.count && b >= a.first_chunk[i] && b < a.first_chunk[i + 1];
      const unsigned long long mask = __ballot(mine);
      if (mask) t = base + __ffsll((long long)mask) - 1;
    }
    if (threadIdx.x == 0) s_t = t;
  }
  __syncthreads();
  const int t = s_t;
  if (t < 0) return;
  const AdamGroup h = a.grp[a.group[t]];
  const int n = a.n[t];
  const int64_t base = (int64_t)((int)blockIdx.x - a.first_chunk[t]) * ADAM_CHUNK;
  float* p = a.p[t];
  const float* g = a.g[t];
  float* m = a.m[t];
  float* v = a.v[t];
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
#pragma unroll
  for (int k = 0; k < ADAM_CHUNK / 1024; ++k) {
    const int64_t i = base + (int64_t)k * 1024 + 4 * (int)threadIdx.x;
    if (i >= n) break;
    if (vec && i + 4 <= n) {
      f32x4 pp = *reinterpret_cast<const f32x4*>(p + i), gg = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + i));
      f32x4 mm = *reinterpret_cast<const f32x4*>(m + i), vv = *reinterpret_cast<const f32x4*>(v + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = pp[e], me = mm[e], ve = vv[e];
        adam_one(pe, gg[e], me, ve, h);
        pp[e] = pe; mm[e] = me; vv[e] = ve;
      }
      *reinterpret_cast<f32x4*>(p + i) = pp;
      *reinterpret_cast<f32x4*>(m + i) = mm;
      *reinterpret_cast<f32x4*>(v + i) = vv;
    } else {
      for (int e = 0; e < 4 && i + e < n; ++e) {
        float pp = p[i + e], mm = m[i + e], vv = v[i + e];
        adam_one(pp, g[i + e], mm, vv, h);
        p[i + e] = pp; m[i + e] = mm; v[i + e] = vv;
      }
    }
  }
}

}  // namespace gims

using namespace gims;

extern "C" int gims_adam_step(const gims_adam_tensor* tensors, int32_t count, const gims_adam_group* groups, int32_t n_groups, void* stream) {
  GIMS_CHECK_ARG(count >= 0 && n_groups >= 0 && n_groups <= ADAM_MAX_GROUPS, "gims_adam_step: %d groups (at most %d)", n_groups, ADAM_MAX_GROUPS);
  if (count == 0) return GIMS_OK;
  GIMS_CHECK_ARG(tensors && groups && n_groups > 0, "gims_adam_step: null table");
  AdamGroup hg[ADAM_MAX_GROUPS];
  for (int k = 0; k < n_groups; ++k) {
    const gims_adam_group& g = groups[k];
    GIMS_CHECK_ARG(g.step >= 1 && g.beta1 >= 0.0 && g.beta1 < 1.0 && g.beta2 >= 0.0 && g.beta2 < 1.0 && g.eps >= 0.0 && g.lr >= 0.0 && g.weight_decay >= 0.0,
                   "gims_adam_step: group %d: step %lld lr %g betas (%g, %g) eps %g weight_decay %g", k, (long long)g.step, g.lr, g.beta1, g.beta2, g.eps,
                   g.weight_decay);
    const double bc1 = 1.0 - pow(g.beta1, (double)g.step), bc2 = 1.0 - pow(g.beta2, (double)g.step);
    hg[k].step_size = (float)(g.lr / bc1);
    hg[k].bc2_sqrt = (float)sqrt(bc2);
    hg[k].beta2 = (float)g.beta2;
    hg[k].eps = (float)g.eps;
    hg[k].wd = (float)g.weight_decay;
    hg[k].one_m_beta1 = (float)(1.0 - g.beta1);
    hg[k].one_m_beta2 = (float)(1.0 - g.beta2);
  }
  for (int k = 0; k < count; ++k) {
    GIMS_CHECK_ARG(tensors[k].n >= 0 && tensors[k].n < ((int64_t)1 << 31), "gims_adam_step: tensor %d has %lld elements", k, (long long)tensors[k].n);
    GIMS_CHECK_ARG(tensors[k].n == 0 || (tensors[k].param && tensors[k].grad && tensors[k].exp_avg && tensors[k].exp_avg_sq), "gims_adam_step: tensor %d: null pointer", k);
    GIMS_CHECK_ARG(tensors[k].group >= 0 && tensors[k].group < n_groups, "gims_adam_step: tensor %d names group %d of %d", k, tensors[k].group, n_groups);
  }
  AdamLaunch a;
  for (int k = 0; k < n_groups; ++k) a.grp[k] = hg[k];
  int k = 0;
  while (k < count) {
    int c = 0, chunks = 0;
    for (; k < count && c < ADAM_MAX_TENSORS; ++k) {
      if (tensors[k].n == 0) continue;
      a.p[c] = tensors[k].param; a.g[c] = tensors[k].grad; a.m[c] = tensors[k].exp_avg; a.v[c] = tensors[k].exp_avg_sq;
      a.n[c] = (int)tensors[k].n; a.group[c] = (unsigned char)tensors[k].group;
      a.first_chunk[c] = chunks;
      chunks += cdiv(tensors[k].n, ADAM_CHUNK);
      ++c;
    }
    if (c == 0) break;
    a.first_chunk[c] = chunks;
    a.count = c;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, a);
    GIMS_LAUNCH_CHECK();
  }
  return GIMS_OK;
}

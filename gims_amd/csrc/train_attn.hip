// Attention of the training step (models/gmatcher.py:35-39 inside forward_train, :309-386, and its reverse pass) without the probability
// matrices: one call per GNN layer and direction for ALL images and heads; the forward in exact f32 on the matrix cores, the reverse pass in
// exact f32 or (default of the training step) in three bf16 passes -- the second half of this file.
//
// What it replaces.  trainstep.py ran, per image and layer, a batched split-bf16 product Q K^T that wrote the 4 x n x m scores, a softmax pass
// over them, a second product with V (+ its split-K fold) -- and kept P (67 MB per image and layer at 2048 keypoints) for a reverse pass of four
// more products and a softmax-backward pass: 292 us of kernels and ~35 host-side tensor / launch operations per image and layer, on a step that
// is bound by its HOST time.  Here the scores exist only as 32 x 32 tiles in registers: the forward keeps the row statistic lse = max + log(sum)
// (4 B per query and head), the reverse pass recomputes P = exp(S - lse) from Q, K and lse.
//
// Arithmetic.  v_mfma_f32_32x32x2_f32: f32 operands, f32 accumulation -- bitwise an fmaf chain over the contraction index, so S is the SAME
// bits in the forward and in both reverse kernels (the 1/sqrt(64) = 2^-3 scale is folded into one operand: exact).  The rate is the f32 vector
// peak (157 TFLOP/s, 1/16 of the bf16 matrix rate): 2.7 x the matrix time of six split-bf16 passes, but nothing is split, nothing is stored
// and the products the step's accuracy hangs on (the message feeds 18 ReLU layers whose near-zero pre-activations decide the gradient error,
// tests/test_trainstep_gpu.py) stay in the reference's own arithmetic.
//
// Layout of a tile product.  All three kernels compute their score tile so that the accumulator layout (lane <-> column, 16 registers <-> rows
// (j & 3) + 8 (j >> 2) + 4 (lane >> 5)) is directly the B operand of the NEXT product: the contraction index of that product is the row index of
// the tile, and a 32x32x2 step consumes exactly one accumulator register (its two lane halves hold the two contraction slots) while the A
// operand -- one f32 per lane -- is read from LDS at the row the register stands for.  No transposes through LDS, no shuffles.
//   forward   S^T[key][q] = K Q^T   (lane <-> query: row max / sum in-lane + one lane^32 exchange);   O^T[d][q] += V^T[d][key] P^T[key][q]
//   dQ        S^T, dP^T[key][q] = V dO^T,  dS^T = P^T (dP^T - D_q);                                    dQ^T[d][q] += K^T[d][key] dS^T[key][q]
//   dK, dV    S[q][key] = Q K^T (lane <-> key), dP[q][key] = dO V^T, dS = P (dP - D_q);  dV^T[d][key] += dO^T[d][q] P[q][key],
//                                                                                         dK^T[d][key] += Q^T[d][q] dS[q][key]
// (the reverse pass is two kernels -- each accumulation stays inside one wave, fixed order, no atomics: 7 tile products instead of 5).
// Work decomposition: a wave owns 32 queries (forward, dQ) or 32 keys (dK / dV); a workgroup = 4 waves shares the streamed operand tiles in LDS;
// the streamed dimension is cut into `splits` ranges (one workgroup each) until ~2 waves per SIMD exist, and a small kernel folds the partial
// results in split order (forward: with the usual max / sum rescaling).
#include "common.h"

#include <stdlib.h>

#include <algorithm>

namespace gims {

constexpr int TA_MAXP = 32;          // problems per launch (the table travels in the kernel arguments)
constexpr int TA_TP = 33;            // pitch of a transposed tile [d][row]
constexpr int TA_SP = 72;            // pitch of a row-major tile [row][d]   (4 x pitch = 32 mod 64: the two lane halves hit disjoint banks)
constexpr int TA_MAX_SPLITS = 8;

struct TaK {
  const float* qkv; int64_t ld; int64_t rows;
  float* o; int64_t ldo; float* lse;
  const float* dout; int64_t lddo; float* dqkv; int64_t lddq;
  float* part;                       // forward: [splits][heads][rows][64] O, then [splits][heads][rows][2] (m, l)
                                     // reverse: [heads][rows] D, then [splits][heads][rows][64] x 3 (dQ, dK, dV)
  int heads, d, splits, nprob;
  float scale;
  gims_train_attn_problem pr[TA_MAXP];
};

__host__ __device__ __forceinline__ int64_t ta_dfloats(int heads, int64_t rows) { return ((int64_t)heads * rows + 3) & ~(int64_t)3; }
__device__ __forceinline__ float ta_other_half(float x) { return __shfl_xor(x, 32, 64); }
__device__ __forceinline__ int ta_row_of(int j, int hf) { return (j & 3) + 8 * (j >> 2) + 4 * hf; }

// a 32 x 64 tile of rows [r0, r0 + 32) (clamped to [0, n - 1]) of a row-major matrix: two float4 per thread of a 256-thread workgroup
__device__ __forceinline__ void ta_fetch(const float* base, int64_t ld, int r0, int n, f32x4 (&reg)[2]) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int idx = threadIdx.x + 256 * r, row = idx >> 4, c4 = idx & 15;
    int gr = r0 + row;
    gr = gr < n ? gr : n - 1;
    reg[r] = *(const f32x4*)(base + (int64_t)gr * ld + 4 * c4);
  }
}
__device__ __forceinline__ void ta_put_t(float* lds, const f32x4 (&reg)[2]) {      // [d][row], pitch 33
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int idx = threadIdx.x + 256 * r, row = idx >> 4, c4 = idx & 15;
#pragma unroll
    for (int c = 0; c < 4; ++c) lds[(4 * c4 + c) * TA_TP + row] = reg[r][c];
  }
}
__device__ __forceinline__ void ta_put_s(float* lds, const f32x4 (&reg)[2]) {      // [row][d], pitch 72
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int idx = threadIdx.x + 256 * r, row = idx >> 4, c4 = idx & 15;
    *(f32x4*)(lds + row * TA_SP + 4 * c4) = reg[r];
  }
}

#define TA_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// The A operands of a chain of 16 products, read from LDS AHEAD of the chain that uses them: one ds_read per product issued right in front of it
// left every pair of products waiting ~120 cycles for LDS behind a 64-cycle product (62 % of the f32 matrix rate); with the reads of the NEXT
// 16 products in flight under the current 16 the pipe only waits for itself.  Stages are fenced with sched_barrier so that the compiler keeps
// the reads where they are written.
//   transposed tile [d][row]: operand i of the chain over d = (2 i + hf, ln)
__device__ __forceinline__ void ta_ops_t(float (&r)[16], const float* lds, int i0, int hf, int ln) {
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = lds[(2 * (i0 + i) + hf) * TA_TP + ln];
}
//   row-major tile [row][d]: operands (row of register j, d = ln) and (.., d = 32 + ln) for j = j0 .. j0 + 7
__device__ __forceinline__ void ta_ops_s(float (&r)[16], const float* lds, int j0, int hf, int ln) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    r[2 * j] = lds[ta_row_of(j0 + j, hf) * TA_SP + ln];
    r[2 * j + 1] = lds[ta_row_of(j0 + j, hf) * TA_SP + 32 + ln];
  }
}
#define TA_FENCE() __builtin_amdgcn_sched_barrier(0)

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void ta_fwd_kernel(TaK a) {
  __shared__ __attribute__((aligned(16))) float kt[64 * TA_TP];
  __shared__ __attribute__((aligned(16))) float vs[32 * TA_SP];
  const gims_train_attn_problem pr = a.pr[blockIdx.z];
  const int q0 = blockIdx.x * 128;
  if (q0 >= pr.nq) return;
  const int h = blockIdx.y / a.splits, sp = blockIdx.y % a.splits;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ln = lane & 31, hf = lane >> 5;
  const int ntile = (pr.nk + 31) >> 5;
  const int t_lo = (int)((int64_t)sp * ntile / a.splits), t_hi = (int)((int64_t)(sp + 1) * ntile / a.splits);
  const int qrow = q0 + wave * 32 + ln;
  const float* kbase = a.qkv + (int64_t)pr.k_off * a.ld + a.d + h * 64;
  const float* vbase = kbase + a.d;
  float qf[32];
  {
    const float* qp = a.qkv + (int64_t)(pr.q_off + (qrow < pr.nq ? qrow : pr.nq - 1)) * a.ld + h * 64;
#pragma unroll
    for (int i = 0; i < 32; ++i) qf[i] = qp[2 * i + hf] * a.scale;
  }
  f32x16 o0, o1;
#pragma unroll
  for (int j = 0; j < 16; ++j) o0[j] = o1[j] = 0.f;
  float m = -INFINITY, l = 0.f;
  f32x4 kr[2], vr[2];
  if (t_lo < t_hi) {
    ta_fetch(kbase, a.ld, t_lo * 32, pr.nk, kr);
    ta_fetch(vbase, a.ld, t_lo * 32, pr.nk, vr);
  }
  for (int tile = t_lo; tile < t_hi; ++tile) {
    __syncthreads();
    ta_put_t(kt, kr);
    ta_put_s(vs, vr);
    __syncthreads();
    if (tile + 1 < t_hi) {
      ta_fetch(kbase, a.ld, (tile + 1) * 32, pr.nk, kr);
      ta_fetch(vbase, a.ld, (tile + 1) * 32, pr.nk, vr);
    }
    float pa[16], pb[16];
    ta_ops_t(pa, kt, 0, hf, ln);
    ta_ops_t(pb, kt, 16, hf, ln);
    TA_FENCE();
    f32x16 s;
#pragma unroll
    for (int j = 0; j < 16; ++j) s[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s = TA_MFMA(pa[i], qf[i], s);
    ta_ops_s(pa, vs, 0, hf, ln);                    // (the rows of V the second product starts with: in flight under the rest of the first)
    TA_FENCE();
#pragma unroll
    for (int i = 0; i < 16; ++i) s = TA_MFMA(pb[i], qf[16 + i], s);
    ta_ops_s(pb, vs, 8, hf, ln);
    TA_FENCE();
    if (tile * 32 + 32 > pr.nk) {
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (tile * 32 + ta_row_of(j, hf) >= pr.nk) s[j] = -INFINITY;
    }
    float mx = s[0];
#pragma unroll
    for (int j = 1; j < 16; ++j) mx = fmaxf(mx, s[j]);
    mx = fmaxf(mx, ta_other_half(mx));
    const float mn = fmaxf(m, mx);                 // finite: every tile holds at least one valid key
    const float alpha = __expf(m - mn);
    float ps = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      s[j] = __expf(s[j] - mn);
      ps += s[j];
    }
    ps += ta_other_half(ps);
    l = l * alpha + ps;
    m = mn;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {       // (the running maximum settles after the first tiles: most tiles rescale nothing)
#pragma unroll
      for (int j = 0; j < 16; ++j) { o0[j] *= alpha; o1[j] *= alpha; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o0 = TA_MFMA(pa[2 * j], s[j], o0);
      o1 = TA_MFMA(pa[2 * j + 1], s[j], o1);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o0 = TA_MFMA(pb[2 * j], s[8 + j], o0);
      o1 = TA_MFMA(pb[2 * j + 1], s[8 + j], o1);
    }
  }
  if (qrow >= pr.nq) return;
  const int64_t row = pr.q_off + qrow;
  if (a.splits == 1) {
    const float inv = 1.f / l;
    float* op = a.o + row * a.ldo + h * 64;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      *(f32x4*)(op + 8 * jj + 4 * hf) = f32x4{o0[4 * jj] * inv, o0[4 * jj + 1] * inv, o0[4 * jj + 2] * inv, o0[4 * jj + 3] * inv};
      *(f32x4*)(op + 32 + 8 * jj + 4 * hf) = f32x4{o1[4 * jj] * inv, o1[4 * jj + 1] * inv, o1[4 * jj + 2] * inv, o1[4 * jj + 3] * inv};
    }
    if (hf == 0) a.lse[(int64_t)h * a.rows + row] = m + logf(l);
  } else {
    const int64_t slot = ((int64_t)sp * a.heads + h) * a.rows + row;
    float* op = a.part + slot * 64;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      *(f32x4*)(op + 8 * jj + 4 * hf) = f32x4{o0[4 * jj], o0[4 * jj + 1], o0[4 * jj + 2], o0[4 * jj + 3]};
      *(f32x4*)(op + 32 + 8 * jj + 4 * hf) = f32x4{o1[4 * jj], o1[4 * jj + 1], o1[4 * jj + 2], o1[4 * jj + 3]};
    }
    if (hf == 0) {
      float* ml = a.part + (int64_t)a.splits * a.heads * a.rows * 64 + slot * 2;
      ml[0] = m;
      ml[1] = l;
    }
  }
}

// fold of the forward partials, in split order: o = sum_s e^(m_s - m) O_s / sum_s e^(m_s - m) l_s
__global__ __launch_bounds__(256) void ta_fwd_merge_kernel(TaK a) {
  const gims_train_attn_problem pr = a.pr[blockIdx.z];
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c4 = (int)(idx & 15), h = (int)((idx >> 4) % a.heads);
  const int64_t q = (idx >> 4) / a.heads;
  if (q >= pr.nq) return;
  const int64_t row = pr.q_off + q;
  const float* ml0 = a.part + (int64_t)a.splits * a.heads * a.rows * 64;
  float ms[TA_MAX_SPLITS], m = -INFINITY;
  for (int s = 0; s < a.splits; ++s) {
    ms[s] = ml0[(((int64_t)s * a.heads + h) * a.rows + row) * 2];
    m = fmaxf(m, ms[s]);
  }
  float L = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < a.splits; ++s) {
    const int64_t slot = ((int64_t)s * a.heads + h) * a.rows + row;
    const float w = __expf(ms[s] - m);                     // an empty split has m_s = -inf, l_s = 0, O_s = 0
    L += w * ml0[slot * 2 + 1];
    const f32x4 os = *(const f32x4*)(a.part + slot * 64 + 4 * c4);
    acc += os * w;
  }
  const float inv = 1.f / L;
  *(f32x4*)(a.o + row * a.ldo + h * 64 + 4 * c4) = acc * inv;
  if (c4 == 0) a.lse[(int64_t)h * a.rows + row] = m + logf(L);
}

// ------------------------------------------------------------------------------------------------ reverse pass: dQ (and D = rowsum(dO * O))
__global__ __launch_bounds__(256) void ta_bwd_q_kernel(TaK a) {
  __shared__ __attribute__((aligned(16))) float kt[64 * TA_TP];
  __shared__ __attribute__((aligned(16))) float vt[64 * TA_TP];
  __shared__ __attribute__((aligned(16))) float ks[32 * TA_SP];
  const gims_train_attn_problem pr = a.pr[blockIdx.z];
  const int q0 = blockIdx.x * 128;
  if (q0 >= pr.nq) return;
  const int h = blockIdx.y / a.splits, sp = blockIdx.y % a.splits;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ln = lane & 31, hf = lane >> 5;
  const int ntile = (pr.nk + 31) >> 5;
  const int t_lo = (int)((int64_t)sp * ntile / a.splits), t_hi = (int)((int64_t)(sp + 1) * ntile / a.splits);
  const int qrow = q0 + wave * 32 + ln;
  const int64_t row = pr.q_off + (qrow < pr.nq ? qrow : pr.nq - 1);
  const float* kbase = a.qkv + (int64_t)pr.k_off * a.ld + a.d + h * 64;
  const float* vbase = kbase + a.d;
  float qf[32], dof[32];
  float dsum = 0.f;
  {
    const float* qp = a.qkv + row * a.ld + h * 64;
    const float* dp = a.dout + row * a.lddo + h * 64;
    const float* op = a.o + row * a.ldo + h * 64;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      qf[i] = qp[2 * i + hf] * a.scale;
      dof[i] = dp[2 * i + hf];
      dsum = fmaf(dof[i], op[2 * i + hf], dsum);
    }
    dsum += ta_other_half(dsum);
  }
  const float lse = a.lse[(int64_t)h * a.rows + row];
  float* dsum_out = a.part;
  if (sp == 0 && hf == 0 && qrow < pr.nq) dsum_out[(int64_t)h * a.rows + row] = dsum;
  f32x16 g0, g1;
#pragma unroll
  for (int j = 0; j < 16; ++j) g0[j] = g1[j] = 0.f;
  f32x4 kr[2], vr[2];
  if (t_lo < t_hi) {
    ta_fetch(kbase, a.ld, t_lo * 32, pr.nk, kr);
    ta_fetch(vbase, a.ld, t_lo * 32, pr.nk, vr);
  }
  for (int tile = t_lo; tile < t_hi; ++tile) {
    __syncthreads();
    ta_put_t(kt, kr);
    ta_put_s(ks, kr);
    ta_put_t(vt, vr);
    __syncthreads();
    if (tile + 1 < t_hi) {
      ta_fetch(kbase, a.ld, (tile + 1) * 32, pr.nk, kr);
      ta_fetch(vbase, a.ld, (tile + 1) * 32, pr.nk, vr);
    }
    float pa[16], pb[16];
    ta_ops_t(pa, kt, 0, hf, ln);
    ta_ops_t(pb, kt, 16, hf, ln);
    TA_FENCE();
    f32x16 s, dp;
#pragma unroll
    for (int j = 0; j < 16; ++j) s[j] = dp[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s = TA_MFMA(pa[i], qf[i], s);
    ta_ops_t(pa, vt, 0, hf, ln);
    TA_FENCE();
#pragma unroll
    for (int i = 0; i < 16; ++i) s = TA_MFMA(pb[i], qf[16 + i], s);
    ta_ops_t(pb, vt, 16, hf, ln);
    TA_FENCE();
#pragma unroll
    for (int i = 0; i < 16; ++i) dp = TA_MFMA(pa[i], dof[i], dp);
    ta_ops_s(pa, ks, 0, hf, ln);
    TA_FENCE();
#pragma unroll
    for (int i = 0; i < 16; ++i) dp = TA_MFMA(pb[i], dof[16 + i], dp);
    ta_ops_s(pb, ks, 8, hf, ln);
    TA_FENCE();
    if (tile * 32 + 32 > pr.nk) {                          // the last, ragged tile only (a uniform branch, not sixteen selects per tile): P = e^-inf = 0
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (tile * 32 + ta_row_of(j, hf) >= pr.nk) s[j] = -INFINITY;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) s[j] = __expf(s[j] - lse) * (dp[j] - dsum) * a.scale;      // dS^T, with the 1/sqrt(d_head) of dQ = dS K / sqrt(d_head)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      g0 = TA_MFMA(pa[2 * j], s[j], g0);
      g1 = TA_MFMA(pa[2 * j + 1], s[j], g1);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      g0 = TA_MFMA(pb[2 * j], s[8 + j], g0);
      g1 = TA_MFMA(pb[2 * j + 1], s[8 + j], g1);
    }
  }
  if (qrow >= pr.nq) return;
  float* gp = a.splits == 1 ? a.dqkv + row * a.lddq + h * 64
                            : a.part + ta_dfloats(a.heads, a.rows) + (((int64_t)sp * a.heads + h) * a.rows + row) * 64;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    *(f32x4*)(gp + 8 * jj + 4 * hf) = f32x4{g0[4 * jj], g0[4 * jj + 1], g0[4 * jj + 2], g0[4 * jj + 3]};
    *(f32x4*)(gp + 32 + 8 * jj + 4 * hf) = f32x4{g1[4 * jj], g1[4 * jj + 1], g1[4 * jj + 2], g1[4 * jj + 3]};
  }
}

// ------------------------------------------------------------------------------------------------ reverse pass: dK, dV
__global__ __launch_bounds__(256, 2) void ta_bwd_kv_kernel(TaK a) {
  __shared__ __attribute__((aligned(16))) float qt[64 * TA_TP];
  __shared__ __attribute__((aligned(16))) float qs[32 * TA_SP];
  __shared__ __attribute__((aligned(16))) float dt[64 * TA_TP];
  __shared__ __attribute__((aligned(16))) float ds_[32 * TA_SP];
  __shared__ __attribute__((aligned(16))) float lse_s[32], dsum_s[32];
  const gims_train_attn_problem pr = a.pr[blockIdx.z];
  const int k0 = blockIdx.x * 128;
  if (k0 >= pr.nk) return;
  const int h = blockIdx.y / a.splits, sp = blockIdx.y % a.splits;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ln = lane & 31, hf = lane >> 5;
  const int ntile = (pr.nq + 31) >> 5;
  const int t_lo = (int)((int64_t)sp * ntile / a.splits), t_hi = (int)((int64_t)(sp + 1) * ntile / a.splits);
  const int krow = k0 + wave * 32 + ln;
  const int64_t row = pr.k_off + (krow < pr.nk ? krow : pr.nk - 1);
  const float* qbase = a.qkv + (int64_t)pr.q_off * a.ld + h * 64;
  const float* dbase = a.dout + (int64_t)pr.q_off * a.lddo + h * 64;
  const float* lbase = a.lse + (int64_t)h * a.rows + pr.q_off;
  const float* sbase = a.part + (int64_t)h * a.rows + pr.q_off;
  float kf[32], vf[32];
  {
    const float* kp = a.qkv + row * a.ld + a.d + h * 64;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      kf[i] = kp[2 * i + hf] * a.scale;
      vf[i] = kp[a.d + 2 * i + hf];
    }
  }
  f32x16 gk0, gk1, gv0, gv1;
#pragma unroll
  for (int j = 0; j < 16; ++j) gk0[j] = gk1[j] = gv0[j] = gv1[j] = 0.f;
  f32x4 qr[2], dr[2];
  float st = 0.f;                                          // threads 0..31: lse of the tile's queries, 32..63: D
  auto fetch_stat = [&](int tile) {
    if (threadIdx.x < 64) {
      int q = tile * 32 + (threadIdx.x & 31);
      q = q < pr.nq ? q : pr.nq - 1;
      st = threadIdx.x < 32 ? lbase[q] : sbase[q];
    }
  };
  if (t_lo < t_hi) {
    ta_fetch(qbase, a.ld, t_lo * 32, pr.nq, qr);
    ta_fetch(dbase, a.lddo, t_lo * 32, pr.nq, dr);
    fetch_stat(t_lo);
  }
  for (int tile = t_lo; tile < t_hi; ++tile) {
    __syncthreads();
    ta_put_t(qt, qr);
    ta_put_s(qs, qr);
    ta_put_t(dt, dr);
    ta_put_s(ds_, dr);
    if (threadIdx.x < 32) lse_s[threadIdx.x] = st;
    else if (threadIdx.x < 64) dsum_s[threadIdx.x - 32] = st;
    __syncthreads();
    if (tile + 1 < t_hi) {
      ta_fetch(qbase, a.ld, (tile + 1) * 32, pr.nq, qr);
      ta_fetch(dbase, a.lddo, (tile + 1) * 32, pr.nq, dr);
      fetch_stat(tile + 1);
    }
    float pa[16], pb[16];
    ta_ops_t(pa, qt, 0, hf, ln);
    ta_ops_t(pb, qt, 16, hf, ln);
    TA_FENCE();
    f32x16 s, dp;
#pragma unroll
    for (int j = 0; j < 16; ++j) s[j] = dp[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s = TA_MFMA(pa[i], kf[i], s);                                   // S[q][key]
    ta_ops_t(pa, dt, 0, hf, ln);
    TA_FENCE();
#pragma unroll
    for (int i = 0; i < 16; ++i) s = TA_MFMA(pb[i], kf[16 + i], s);
    ta_ops_t(pb, dt, 16, hf, ln);
    TA_FENCE();
#pragma unroll
    for (int i = 0; i < 16; ++i) dp = TA_MFMA(pa[i], vf[i], dp);                                 // dP[q][key]
    ta_ops_s(pa, ds_, 0, hf, ln);
    TA_FENCE();
#pragma unroll
    for (int i = 0; i < 16; ++i) dp = TA_MFMA(pb[i], vf[16 + i], dp);
    ta_ops_s(pb, ds_, 8, hf, ln);
    TA_FENCE();
    if (tile * 32 + 32 > pr.nq) {                          // the last, ragged tile only: P = e^-inf = 0 for the rows past the queries
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (tile * 32 + ta_row_of(j, hf) >= pr.nq) s[j] = -INFINITY;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int q = ta_row_of(j, hf);
      const float p = __expf(s[j] - lse_s[q]);
      s[j] = p;
      dp[j] = p * (dp[j] - dsum_s[q]);                     // dS[q][key]
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      gv0 = TA_MFMA(pa[2 * j], s[j], gv0);
      gv1 = TA_MFMA(pa[2 * j + 1], s[j], gv1);
    }
    ta_ops_s(pa, qs, 0, hf, ln);
    TA_FENCE();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      gv0 = TA_MFMA(pb[2 * j], s[8 + j], gv0);
      gv1 = TA_MFMA(pb[2 * j + 1], s[8 + j], gv1);
    }
    ta_ops_s(pb, qs, 8, hf, ln);
    TA_FENCE();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      gk0 = TA_MFMA(pa[2 * j], dp[j], gk0);
      gk1 = TA_MFMA(pa[2 * j + 1], dp[j], gk1);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      gk0 = TA_MFMA(pb[2 * j], dp[8 + j], gk0);
      gk1 = TA_MFMA(pb[2 * j + 1], dp[8 + j], gk1);
    }
  }
  if (krow >= pr.nk) return;
  float *kp, *vp;
  if (a.splits == 1) {
    kp = a.dqkv + row * a.lddq + a.d + h * 64;
    vp = kp + a.d;
  } else {
    const int64_t blk = (int64_t)a.splits * a.heads * a.rows * 64;
    kp = a.part + ta_dfloats(a.heads, a.rows) + blk + (((int64_t)sp * a.heads + h) * a.rows + row) * 64;
    vp = kp + blk;
  }
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    *(f32x4*)(kp + 8 * jj + 4 * hf) = f32x4{gk0[4 * jj], gk0[4 * jj + 1], gk0[4 * jj + 2], gk0[4 * jj + 3]} * a.scale;
    *(f32x4*)(kp + 32 + 8 * jj + 4 * hf) = f32x4{gk1[4 * jj], gk1[4 * jj + 1], gk1[4 * jj + 2], gk1[4 * jj + 3]} * a.scale;
    *(f32x4*)(vp + 8 * jj + 4 * hf) = f32x4{gv0[4 * jj], gv0[4 * jj + 1], gv0[4 * jj + 2], gv0[4 * jj + 3]};
    *(f32x4*)(vp + 32 + 8 * jj + 4 * hf) = f32x4{gv1[4 * jj], gv1[4 * jj + 1], gv1[4 * jj + 2], gv1[4 * jj + 3]};
  }
}

// ------------------------------------------------------------------------------------------------ reverse pass in three bf16 passes
// The same two kernels on v_mfma_f32_32x32x16_bf16 (32 cycles per 32 x 32 x 16 step instead of 64 per 32 x 32 x 2): every operand is split into
// hi = bf16(x), lo = bf16(x - hi) and a product is lo*hi + hi*lo + hi*hi (small terms first).  The reverse pass is LINEAR in its operands -- no
// ReLU decides anything there -- and on every trainstep_* fixture the gradient errors with 16-bit-mantissa reverse products equal those of the
// exact ones to three digits (trainstep.py:_backward_precision).  Tile products per 32 x 32 tile: 12 steps instead of 32, i.e. 384 matrix
// cycles instead of 2048.
// Streamed tiles sit in LDS ROW-MAJOR as two bf16 images [32][64] (pitch 72 elements = 144 B: sixteen lanes reading 16 B of sixteen rows touch
// every bank once).  A tile is read two ways: as the A operand of a product over d (lane <-> row, 8 consecutive d: one ds_read_b128) and, for
// the products that contract over the tile's ROWS, transposed by ds_read_b64_tr_b16 (lane <-> d, k-slots = rows 16 s + 8 (j >> 2) + 4 (lane >> 5)
// + (j & 3): exactly the rows the accumulator registers 8 s .. 8 s + 7 of the score tile stand for -- the addressing of attention.hip's V^T
// fragments, tools/probes/tr16_probe.hip).
constexpr int TB_P = 72;
typedef short s16x4t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float tb_hi_lo16(uint32_t pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float tb_hi_hi16(uint32_t pk) { return __uint_as_float(pk & 0xffff0000u); }
// two f32 -> packed bf16 hi pair and packed bf16 lo pair
__device__ __forceinline__ void tb_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  hi = pack_bf2(x0, x1);
  lo = pack_bf2(x0 - tb_hi_lo16(hi), x1 - tb_hi_hi16(hi));
}
__device__ __forceinline__ void tb_frag8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
  uint4 h, l;
  tb_split2(x[0], x[1], h.x, l.x);
  tb_split2(x[2], x[3], h.y, l.y);
  tb_split2(x[4], x[5], h.z, l.z);
  tb_split2(x[6], x[7], h.w, l.w);
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}
// registers 8 s .. 8 s + 7 of a score tile as the B operand of step s of a product over the tile's rows
__device__ __forceinline__ void tb_frag_acc(const f32x16& x, int s, bf16x8& hi, bf16x8& lo) {
  const float v[8] = {x[8 * s], x[8 * s + 1], x[8 * s + 2], x[8 * s + 3], x[8 * s + 4], x[8 * s + 5], x[8 * s + 6], x[8 * s + 7]};
  tb_frag8(v, hi, lo);
}
__device__ __forceinline__ void tb_put(uint16_t* hi_t, uint16_t* lo_t, const f32x4 (&reg)[2]) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int idx = threadIdx.x + 256 * r, row = idx >> 4, c4 = idx & 15;
    uint2 h, l;
    tb_split2(reg[r][0], reg[r][1], h.x, l.x);
    tb_split2(reg[r][2], reg[r][3], h.y, l.y);
    *(uint2*)(hi_t + row * TB_P + 4 * c4) = h;
    *(uint2*)(lo_t + row * TB_P + 4 * c4) = l;
  }
}
__device__ __forceinline__ bf16x8 tb_row_op(const uint16_t* tile, int st, int ln, int hf) { return *(const bf16x8*)(tile + ln * TB_P + 16 * st + 8 * hf); }
__device__ __forceinline__ bf16x8 tb_tr_op(const uint16_t* tile, int s, int i, int lane) {
  const uint16_t* p = tile + (4 * (lane >> 5) + ((lane & 15) >> 2)) * TB_P + 16 * ((lane >> 4) & 1) + 4 * (lane & 3) + 16 * s * TB_P + 32 * i;
  const s16x4t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4t __attribute__((address_space(3)))*)(p));
  const s16x4t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4t __attribute__((address_space(3)))*)(p + 8 * TB_P));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
}
#define TB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
// c += (ah + al)(bh + bl) without the lo * lo term, small terms first
#define TB_MFMA3(ah, al, bh, bl, c) \
  do {                              \
    (c) = TB_MFMA((al), (bh), (c)); \
    (c) = TB_MFMA((ah), (bl), (c)); \
    (c) = TB_MFMA((ah), (bh), (c)); \
  } while (0)

// 8 consecutive floats of a row as a resident B fragment (optionally scaled by a power of two)
__device__ __forceinline__ void tb_load8(const float* p, float scale, float (&x)[8]) {
  const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) { x[e] = a[e] * scale; x[4 + e] = b[e] * scale; }
}

__global__ __launch_bounds__(256, 2) void tb_bwd_q_kernel(TaK a) {
  __shared__ __attribute__((aligned(16))) uint16_t kh[32 * TB_P], kl[32 * TB_P], vh[32 * TB_P], vl[32 * TB_P];
  const gims_train_attn_problem pr = a.pr[blockIdx.z];
  const int q0 = blockIdx.x * 128;
  if (q0 >= pr.nq) return;
  const int h = blockIdx.y / a.splits, sp = blockIdx.y % a.splits;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ln = lane & 31, hf = lane >> 5;
  const int ntile = (pr.nk + 31) >> 5;
  const int t_lo = (int)((int64_t)sp * ntile / a.splits), t_hi = (int)((int64_t)(sp + 1) * ntile / a.splits);
  const int qrow = q0 + wave * 32 + ln;
  const int64_t row = pr.q_off + (qrow < pr.nq ? qrow : pr.nq - 1);
  const float* kbase = a.qkv + (int64_t)pr.k_off * a.ld + a.d + h * 64;
  const float* vbase = kbase + a.d;
  bf16x8 qhi[4], qlo[4], dohi[4], dolo[4];
  float dsum = 0.f;
  {
    const float* qp = a.qkv + row * a.ld + h * 64 + 8 * hf;
    const float* dp = a.dout + row * a.lddo + h * 64 + 8 * hf;
    const float* op = a.o + row * a.ldo + h * 64 + 8 * hf;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      float q8[8], d8[8], o8[8];
      tb_load8(qp + 16 * st, a.scale, q8);
      tb_load8(dp + 16 * st, 1.f, d8);
      tb_load8(op + 16 * st, 1.f, o8);
#pragma unroll
      for (int e = 0; e < 8; ++e) dsum = fmaf(d8[e], o8[e], dsum);
      tb_frag8(q8, qhi[st], qlo[st]);
      tb_frag8(d8, dohi[st], dolo[st]);
    }
    dsum += ta_other_half(dsum);
  }
  const float lse = a.lse[(int64_t)h * a.rows + row];
  if (sp == 0 && hf == 0 && qrow < pr.nq) a.part[(int64_t)h * a.rows + row] = dsum;
  f32x16 g0, g1;
#pragma unroll
  for (int j = 0; j < 16; ++j) g0[j] = g1[j] = 0.f;
  f32x4 kr[2], vr[2];
  if (t_lo < t_hi) {
    ta_fetch(kbase, a.ld, t_lo * 32, pr.nk, kr);
    ta_fetch(vbase, a.ld, t_lo * 32, pr.nk, vr);
  }
  for (int tile = t_lo; tile < t_hi; ++tile) {
    __syncthreads();
    tb_put(kh, kl, kr);
    tb_put(vh, vl, vr);
    __syncthreads();
    if (tile + 1 < t_hi) {
      ta_fetch(kbase, a.ld, (tile + 1) * 32, pr.nk, kr);
      ta_fetch(vbase, a.ld, (tile + 1) * 32, pr.nk, vr);
    }
    // Operand fragments are read ONE product group ahead of the group that multiplies them (two register sets, fenced: left to the compiler
    // every group was "4 LDS reads, wait for all of them, 3 MFMAs" -- the matrix pipe idle for an LDS round trip per 96 cycles of work).
    f32x16 s, dp;
#pragma unroll
    for (int j = 0; j < 16; ++j) s[j] = dp[j] = 0.f;
    bf16x8 xh, xl, yh, yl;
    xh = tb_row_op(kh, 0, ln, hf); xl = tb_row_op(kl, 0, ln, hf);
#pragma unroll
    for (int st = 0; st < 4; ++st) {                       // S^T[key][q], dP^T[key][q]
      yh = tb_row_op(vh, st, ln, hf); yl = tb_row_op(vl, st, ln, hf);
      TA_FENCE();
      TB_MFMA3(xh, xl, qhi[st], qlo[st], s);
      TA_FENCE();
      if (st < 3) { xh = tb_row_op(kh, st + 1, ln, hf); xl = tb_row_op(kl, st + 1, ln, hf); }
      else { xh = tb_tr_op(kh, 0, 0, lane); xl = tb_tr_op(kl, 0, 0, lane); }       // the first fragments of the product over the keys
      TA_FENCE();
      TB_MFMA3(yh, yl, dohi[st], dolo[st], dp);
      TA_FENCE();
    }
    if (tile * 32 + 32 > pr.nk) {                          // the last, ragged tile only (a uniform branch, not sixteen selects per tile): P = e^-inf = 0
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (tile * 32 + ta_row_of(j, hf) >= pr.nk) s[j] = -INFINITY;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) s[j] = __expf(s[j] - lse) * (dp[j] - dsum) * a.scale;      // dS^T, with the 1/sqrt(d_head) of dQ = dS K / sqrt(d_head)
    {
      bf16x8 sh, sl;
      tb_frag_acc(s, 0, sh, sl);
      yh = tb_tr_op(kh, 0, 1, lane); yl = tb_tr_op(kl, 0, 1, lane);
      TA_FENCE();
      TB_MFMA3(xh, xl, sh, sl, g0);
      TA_FENCE();
      xh = tb_tr_op(kh, 1, 0, lane); xl = tb_tr_op(kl, 1, 0, lane);
      TA_FENCE();
      TB_MFMA3(yh, yl, sh, sl, g1);
      TA_FENCE();
      tb_frag_acc(s, 1, sh, sl);
      yh = tb_tr_op(kh, 1, 1, lane); yl = tb_tr_op(kl, 1, 1, lane);
      TA_FENCE();
      TB_MFMA3(xh, xl, sh, sl, g0);
      TA_FENCE();
      TB_MFMA3(yh, yl, sh, sl, g1);
    }
  }
  if (qrow >= pr.nq) return;
  float* gp = a.splits == 1 ? a.dqkv + row * a.lddq + h * 64
                            : a.part + ta_dfloats(a.heads, a.rows) + (((int64_t)sp * a.heads + h) * a.rows + row) * 64;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    *(f32x4*)(gp + 8 * jj + 4 * hf) = f32x4{g0[4 * jj], g0[4 * jj + 1], g0[4 * jj + 2], g0[4 * jj + 3]};
    *(f32x4*)(gp + 32 + 8 * jj + 4 * hf) = f32x4{g1[4 * jj], g1[4 * jj + 1], g1[4 * jj + 2], g1[4 * jj + 3]};
  }
}

__global__ __launch_bounds__(256, 2) void tb_bwd_kv_kernel(TaK a) {
  __shared__ __attribute__((aligned(16))) uint16_t qh[32 * TB_P], ql[32 * TB_P], dh[32 * TB_P], dl[32 * TB_P];
  __shared__ float lse_s[32], dsum_s[32];
  const gims_train_attn_problem pr = a.pr[blockIdx.z];
  const int k0 = blockIdx.x * 128;
  if (k0 >= pr.nk) return;
  const int h = blockIdx.y / a.splits, sp = blockIdx.y % a.splits;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ln = lane & 31, hf = lane >> 5;
  const int ntile = (pr.nq + 31) >> 5;
  const int t_lo = (int)((int64_t)sp * ntile / a.splits), t_hi = (int)((int64_t)(sp + 1) * ntile / a.splits);
  const int krow = k0 + wave * 32 + ln;
  const int64_t row = pr.k_off + (krow < pr.nk ? krow : pr.nk - 1);
  const float* qbase = a.qkv + (int64_t)pr.q_off * a.ld + h * 64;
  const float* dbase = a.dout + (int64_t)pr.q_off * a.lddo + h * 64;
  const float* lbase = a.lse + (int64_t)h * a.rows + pr.q_off;
  const float* sbase = a.part + (int64_t)h * a.rows + pr.q_off;
  bf16x8 khi[4], klo[4], vhi[4], vlo[4];
  {
    const float* kp = a.qkv + row * a.ld + a.d + h * 64 + 8 * hf;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      float k8[8], v8[8];
      tb_load8(kp + 16 * st, a.scale, k8);
      tb_load8(kp + a.d + 16 * st, 1.f, v8);
      tb_frag8(k8, khi[st], klo[st]);
      tb_frag8(v8, vhi[st], vlo[st]);
    }
  }
  f32x16 gk0, gk1, gv0, gv1;
#pragma unroll
  for (int j = 0; j < 16; ++j) gk0[j] = gk1[j] = gv0[j] = gv1[j] = 0.f;
  f32x4 qr[2], dr[2];
  float st_ = 0.f;                                         // threads 0..31: lse of the tile's queries, 32..63: D
  auto fetch_stat = [&](int tile) {
    if (threadIdx.x < 64) {
      int q = tile * 32 + (threadIdx.x & 31);
      q = q < pr.nq ? q : pr.nq - 1;
      st_ = threadIdx.x < 32 ? lbase[q] : sbase[q];
    }
  };
  if (t_lo < t_hi) {
    ta_fetch(qbase, a.ld, t_lo * 32, pr.nq, qr);
    ta_fetch(dbase, a.lddo, t_lo * 32, pr.nq, dr);
    fetch_stat(t_lo);
  }
  for (int tile = t_lo; tile < t_hi; ++tile) {
    __syncthreads();
    tb_put(qh, ql, qr);
    tb_put(dh, dl, dr);
    if (threadIdx.x < 32) lse_s[threadIdx.x] = st_;
    else if (threadIdx.x < 64) dsum_s[threadIdx.x - 32] = st_;
    __syncthreads();
    if (tile + 1 < t_hi) {
      ta_fetch(qbase, a.ld, (tile + 1) * 32, pr.nq, qr);
      ta_fetch(dbase, a.lddo, (tile + 1) * 32, pr.nq, dr);
      fetch_stat(tile + 1);
    }
    // (operand fragments one product group ahead, like tb_bwd_q_kernel)
    f32x16 s, dp;
#pragma unroll
    for (int j = 0; j < 16; ++j) s[j] = dp[j] = 0.f;
    bf16x8 xh, xl, yh, yl;
    xh = tb_row_op(qh, 0, ln, hf); xl = tb_row_op(ql, 0, ln, hf);
#pragma unroll
    for (int st = 0; st < 4; ++st) {                       // S[q][key], dP[q][key]
      yh = tb_row_op(dh, st, ln, hf); yl = tb_row_op(dl, st, ln, hf);
      TA_FENCE();
      TB_MFMA3(xh, xl, khi[st], klo[st], s);
      TA_FENCE();
      if (st < 3) { xh = tb_row_op(qh, st + 1, ln, hf); xl = tb_row_op(ql, st + 1, ln, hf); }
      else { xh = tb_tr_op(dh, 0, 0, lane); xl = tb_tr_op(dl, 0, 0, lane); }       // the first fragments of the products over the queries
      TA_FENCE();
      TB_MFMA3(yh, yl, vhi[st], vlo[st], dp);
      TA_FENCE();
    }
    if (tile * 32 + 32 > pr.nq) {                          // the last, ragged tile only: P = e^-inf = 0 for the rows past the queries
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (tile * 32 + ta_row_of(j, hf) >= pr.nq) s[j] = -INFINITY;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int q = ta_row_of(j, hf);
      const float p = __expf(s[j] - lse_s[q]);
      s[j] = p;
      dp[j] = p * (dp[j] - dsum_s[q]);                     // dS[q][key]
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 ph, pl, sh, sl;
      tb_frag_acc(s, s2, ph, pl);
      tb_frag_acc(dp, s2, sh, sl);
      yh = tb_tr_op(qh, s2, 0, lane); yl = tb_tr_op(ql, s2, 0, lane);
      TA_FENCE();
      TB_MFMA3(xh, xl, ph, pl, gv0);                       // dV^T[d][key] += dO^T[d][q] P[q][key]
      TA_FENCE();
      xh = tb_tr_op(dh, s2, 1, lane); xl = tb_tr_op(dl, s2, 1, lane);
      TA_FENCE();
      TB_MFMA3(yh, yl, sh, sl, gk0);                       // dK^T[d][key] += Q^T[d][q] dS[q][key]
      TA_FENCE();
      yh = tb_tr_op(qh, s2, 1, lane); yl = tb_tr_op(ql, s2, 1, lane);
      TA_FENCE();
      TB_MFMA3(xh, xl, ph, pl, gv1);
      TA_FENCE();
      if (s2 == 0) { xh = tb_tr_op(dh, 1, 0, lane); xl = tb_tr_op(dl, 1, 0, lane); }
      TA_FENCE();
      TB_MFMA3(yh, yl, sh, sl, gk1);
      TA_FENCE();
    }
  }
  if (krow >= pr.nk) return;
  float *kp, *vp;
  if (a.splits == 1) {
    kp = a.dqkv + row * a.lddq + a.d + h * 64;
    vp = kp + a.d;
  } else {
    const int64_t blk = (int64_t)a.splits * a.heads * a.rows * 64;
    kp = a.part + ta_dfloats(a.heads, a.rows) + blk + (((int64_t)sp * a.heads + h) * a.rows + row) * 64;
    vp = kp + blk;
  }
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    *(f32x4*)(kp + 8 * jj + 4 * hf) = f32x4{gk0[4 * jj], gk0[4 * jj + 1], gk0[4 * jj + 2], gk0[4 * jj + 3]} * a.scale;
    *(f32x4*)(kp + 32 + 8 * jj + 4 * hf) = f32x4{gk1[4 * jj], gk1[4 * jj + 1], gk1[4 * jj + 2], gk1[4 * jj + 3]} * a.scale;
    *(f32x4*)(vp + 8 * jj + 4 * hf) = f32x4{gv0[4 * jj], gv0[4 * jj + 1], gv0[4 * jj + 2], gv0[4 * jj + 3]};
    *(f32x4*)(vp + 32 + 8 * jj + 4 * hf) = f32x4{gv1[4 * jj], gv1[4 * jj + 1], gv1[4 * jj + 2], gv1[4 * jj + 3]};
  }
}

// fold of the reverse partials, in split order: which = 0 (dQ, over the problem's query rows), 1 / 2 (dK / dV, over its source rows)
__global__ __launch_bounds__(256) void ta_bwd_merge_kernel(TaK a) {
  const gims_train_attn_problem pr = a.pr[blockIdx.z];
  const int which = blockIdx.y;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c4 = (int)(idx & 15), h = (int)((idx >> 4) % a.heads);
  const int64_t r = (idx >> 4) / a.heads;
  if (r >= (which == 0 ? pr.nq : pr.nk)) return;
  const int64_t row = (which == 0 ? pr.q_off : pr.k_off) + r;
  const int64_t blk = (int64_t)a.splits * a.heads * a.rows * 64;
  const float* src = a.part + ta_dfloats(a.heads, a.rows) + which * blk;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < a.splits; ++s) acc += *(const f32x4*)(src + (((int64_t)s * a.heads + h) * a.rows + row) * 64 + 4 * c4);
  *(f32x4*)(a.dqkv + row * a.lddq + which * a.d + h * 64 + 4 * c4) = acc;
}

// ------------------------------------------------------------------------------------------------ host side
static int ta_splits(const gims_train_attn_args* g, bool over_keys) {
  // enough workgroups for ~2 waves per SIMD (2048 waves), never more ranges than 32-row tiles of the streamed dimension
  int64_t waves = 0;
  int min_tiles = 1 << 30;
  for (int i = 0; i < g->n_problems; ++i) {
    const gims_train_attn_problem& p = g->problems[i];
    waves += (int64_t)g->heads * 4 * cdiv(over_keys ? p.nq : p.nk, 128);
    min_tiles = std::min(min_tiles, cdiv(over_keys ? p.nk : p.nq, 32));
  }
  int s = (int)std::min<int64_t>(TA_MAX_SPLITS, std::max<int64_t>(1, (2048 + waves - 1) / std::max<int64_t>(waves, 1)));
  const char* e = getenv("GIMS_TRAIN_ATTN_SPLITS");
  if (e && atoi(e) >= 1) s = std::min(atoi(e), TA_MAX_SPLITS);
  return std::max(1, std::min(s, min_tiles));
}

static int ta_check(const gims_train_attn_args* g, const char* who, bool reverse) {
  GIMS_CHECK_ARG(g && g->qkv && g->o && g->lse && g->problems, "%s: null pointer", who);
  GIMS_CHECK_ARG(g->heads >= 1 && g->d == 64 * g->heads, "%s: d = %d, heads = %d (the head dimension is 64)", who, g->d, g->heads);
  GIMS_CHECK_ARG(g->ld >= 3 * g->d && g->ldo >= g->d && (g->ld & 3) == 0 && (g->ldo & 3) == 0 && ((uintptr_t)g->qkv & 15) == 0 && ((uintptr_t)g->o & 15) == 0,
                 "%s: pitches must cover 3 d / d floats and be multiples of 4, pointers 16-byte aligned", who);
  GIMS_CHECK_ARG(g->n_problems >= 1 && g->rows >= 1, "%s: no problems", who);
  GIMS_CHECK_ARG(!g->work || ((uintptr_t)g->work & 15) == 0, "%s: the workspace must be 16-byte aligned", who);
  for (int i = 0; i < g->n_problems; ++i) {
    const gims_train_attn_problem& p = g->problems[i];
    GIMS_CHECK_ARG(p.nq >= 1 && p.nk >= 1 && p.q_off >= 0 && p.k_off >= 0 && (int64_t)p.q_off + p.nq <= g->rows && (int64_t)p.k_off + p.nk <= g->rows,
                   "%s: problem %d (queries %d + %d, sources %d + %d) does not lie inside the %lld rows", who, i, p.q_off, p.nq, p.k_off, p.nk, (long long)g->rows);
  }
  if (reverse)
    GIMS_CHECK_ARG(g->reverse_precision == GIMS_TRAIN_ATTN_REVERSE_F32 || g->reverse_precision == GIMS_TRAIN_ATTN_REVERSE_BF16X3,
                   "%s: unknown reverse_precision %d", who, g->reverse_precision);
  if (reverse)
    GIMS_CHECK_ARG(g->d_o && g->d_qkv && g->lddo >= g->d && g->lddq >= 3 * g->d && (g->lddo & 3) == 0 && (g->lddq & 3) == 0 && ((uintptr_t)g->d_o & 15) == 0 &&
                       ((uintptr_t)g->d_qkv & 15) == 0,
                   "%s: gradient tensors (pitches multiples of 4, 16-byte aligned)", who);
  return GIMS_OK;
}

static TaK ta_args(const gims_train_attn_args* g, int first, int count, int splits) {
  TaK k{};
  k.qkv = g->qkv; k.ld = g->ld; k.rows = g->rows; k.o = g->o; k.ldo = g->ldo; k.lse = g->lse;
  k.dout = g->d_o; k.lddo = g->lddo; k.dqkv = g->d_qkv; k.lddq = g->lddq; k.part = g->work;
  k.heads = g->heads; k.d = g->d; k.splits = splits; k.nprob = count; k.scale = g->scale;
  for (int i = 0; i < count; ++i) k.pr[i] = g->problems[first + i];
  return k;
}

}  // namespace gims

using namespace gims;

extern "C" size_t gims_train_attention_workspace_floats(int64_t rows, int32_t heads) {
  if (rows <= 0 || heads <= 0) return 0;
  return (size_t)ta_dfloats(heads, rows) + (size_t)heads * (size_t)rows * (size_t)TA_MAX_SPLITS * 64 * 3;
}

extern "C" int gims_train_attention_forward(const gims_train_attn_args* g, void* stream) {
  if (int rc = ta_check(g, "gims_train_attention_forward", false)) return rc;
  const int splits = ta_splits(g, true);
  GIMS_CHECK_ARG(splits == 1 || (g->work && g->work_floats >= gims_train_attention_workspace_floats(g->rows, g->heads)),
                 "gims_train_attention_forward: workspace too small (gims_train_attention_workspace_floats)");
  hipStream_t s = (hipStream_t)stream;
  for (int first = 0; first < g->n_problems; first += TA_MAXP) {
    const int count = std::min(TA_MAXP, g->n_problems - first);
    const TaK k = ta_args(g, first, count, splits);
    int maxq = 0;
    for (int i = 0; i < count; ++i) maxq = std::max(maxq, k.pr[i].nq);
    hipLaunchKernelGGL(ta_fwd_kernel, dim3(cdiv(maxq, 128), g->heads * splits, count), dim3(256), 0, s, k);
    GIMS_LAUNCH_CHECK();
    if (splits > 1) {
      hipLaunchKernelGGL(ta_fwd_merge_kernel, dim3(cdiv((int64_t)maxq * g->heads * 16, 256), 1, count), dim3(256), 0, s, k);
      GIMS_LAUNCH_CHECK();
    }
  }
  return GIMS_OK;
}

extern "C" int gims_train_attention_backward(const gims_train_attn_args* g, void* stream) {
  if (int rc = ta_check(g, "gims_train_attention_backward", true)) return rc;
  GIMS_CHECK_ARG(g->work && g->work_floats >= gims_train_attention_workspace_floats(g->rows, g->heads),
                 "gims_train_attention_backward: workspace too small (gims_train_attention_workspace_floats)");
  // ONE split count for both kernels (they share the partial layout); a problem's two dimensions differ by the kept counts of its two images
  const int splits = std::min(ta_splits(g, true), ta_splits(g, false));
  hipStream_t s = (hipStream_t)stream;
  for (int first = 0; first < g->n_problems; first += TA_MAXP) {
    const int count = std::min(TA_MAXP, g->n_problems - first);
    const TaK k = ta_args(g, first, count, splits);
    int maxq = 0, maxk = 0;
    for (int i = 0; i < count; ++i) {
      maxq = std::max(maxq, k.pr[i].nq);
      maxk = std::max(maxk, k.pr[i].nk);
    }
    if (g->reverse_precision == GIMS_TRAIN_ATTN_REVERSE_F32) {
      hipLaunchKernelGGL(ta_bwd_q_kernel, dim3(cdiv(maxq, 128), g->heads * splits, count), dim3(256), 0, s, k);
      GIMS_LAUNCH_CHECK();
      hipLaunchKernelGGL(ta_bwd_kv_kernel, dim3(cdiv(maxk, 128), g->heads * splits, count), dim3(256), 0, s, k);
      GIMS_LAUNCH_CHECK();
    } else {
      hipLaunchKernelGGL(tb_bwd_q_kernel, dim3(cdiv(maxq, 128), g->heads * splits, count), dim3(256), 0, s, k);
      GIMS_LAUNCH_CHECK();
      hipLaunchKernelGGL(tb_bwd_kv_kernel, dim3(cdiv(maxk, 128), g->heads * splits, count), dim3(256), 0, s, k);
      GIMS_LAUNCH_CHECK();
    }
    if (splits > 1) {
      hipLaunchKernelGGL(ta_bwd_merge_kernel, dim3(cdiv((int64_t)std::max(maxq, maxk) * g->heads * 16, 256), 3, count), dim3(256), 0, s, k);
      GIMS_LAUNCH_CHECK();
    }
  }
  return GIMS_OK;
}

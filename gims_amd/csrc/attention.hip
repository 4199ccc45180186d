// Flash-style multi-head attention on bf16 MFMA (v_mfma_f32_32x32x16_bf16), head dim 64.
//
//   O[q, h*64+d] = sum_k softmax_k( Q[q,h,:] . K[k,h,:] / 8 ) V[k,h,d]        (gmatcher.py:35-39,109-113)
//
// The N x M score / probability matrices of the reference (two (1,4,N,M) f32 temporaries) are never
// materialised: each wave owns 32 queries and walks the keys in tiles of 64 with an online softmax.
//
// Structure (CDNA4 idioms):
//   * swapped QK^T: S^T = K Q^T, so the MFMA C layout puts ONE query per lane column (col = lane&31) and
//     32 of the tile's 64 keys in that lane's registers -- row max / row sum are in-lane plus a single
//     exchange with lane^32, and the O^T accumulator rescale is lane-local;
//   * P never leaves registers: the C-layout of S^T already is a valid B-operand layout for
//     O^T += V^T P^T once the k-index of that MFMA is *defined* as the key order the lane holds
//     (keys {0-3, 8-11} for lanes 0-31, {4-7, 12-15} for lanes 32-63 of each 16-key step) and V^T is
//     read from LDS in the same order (two ds_read_b64 per fragment) -- no permlane / bpermute;
//   * K tile in LDS row-major with XOR-swizzled 16-byte chunks (conflict-free ds_read_b128); 4-wave kernel: V tile
//     transposed on the way into LDS ([d][key], pitch 68 elements: conflict-free ds_read_b64); 8-wave kernel: see there;
//   * global->register prefetch of the next K/V tile is issued before the MFMAs of the current one and written
//     to the OTHER LDS buffer after them: one barrier per key tile;
//   * the running max is deferred (rescale threshold): the O^T accumulator is only rescaled when a row's max
//     grows by more than 2^5 in probability units -- after the first tiles the accumulators stay in place.
#include "common.h"

#include <atomic>
#include <stdio.h>
#include <stdlib.h>

namespace gims {

constexpr int DH = 64;          // head dim
constexpr int KB = 64;          // keys per tile
constexpr int QW = 32;          // queries per wave
constexpr int ATT_WAVES = 4;    // waves per workgroup
constexpr int QB = QW * ATT_WAVES;
constexpr int VR_LD = 80;       // pitch of a ROW-MAJOR V tile [key][d] in bf16 elements (160 bytes): filled by LDS-DMA, read by ds_read_b64_tr_b16
typedef short s16x4 __attribute__((ext_vector_type(4)));

// The 16-bit operand format of a kernel instance: bf16 (v_mfma_f32_32x32x16_bf16) or, F16, IEEE half (v_mfma_f32_32x32x16_f16: the
// same rate, three more mantissa bits, range 65504).  Fragments travel as 8 x 16-bit lanes either way (LDS images, DMA and the
// transposing reads are type-agnostic); only the MFMA opcode and the f32 -> 16-bit packing of P differ (both round to nearest even).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <bool F16>
__device__ __forceinline__ f32x16 mfma_16b(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ uint32_t pack_16b(float lo, float hi) {
  if constexpr (F16) {
    const f32x2 f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, f16x2));     // v_cvt_pk_f16_f32
  } else return pack_bf2(lo, hi);
}

__device__ __forceinline__ int k_off(int row, int chunk) {  // K tile: [64 keys][64 d], 16-byte chunks swizzled
  return row * DH + ((chunk ^ ((row >> 1) & 7)) << 3);
}

// Peakedness statistic of the softmax rows a wave just finished (attention_precision='auto' of the Python shell decides per
// layer between this file's bf16 kernels and attention_x3_kernel from it): per head, the sum over the reported queries of
// the row maximum of P in 2^-24 fixed point, the number of reported queries, the largest row maximum, and the number of reported
// queries whose row maximum exceeds 1/2 (the TAIL of the distribution: a head with a few one-hot rows among many diffuse ones has a
// small mean).  Integer atomics: the totals do not depend on the arrival order.  stat: [n_heads + 1][4] uint64 {sum, count, max, tail};
// row n_heads belongs to attention_range_kernel.
__device__ __forceinline__ void emit_peak_stat(unsigned long long* stat, int head, float pmax, bool valid) {
  const unsigned int fx = valid ? (unsigned int)(fminf(fmaxf(pmax, 0.f), 1.f) * 16777216.f + 0.5f) : 0u;
  unsigned int sum = fx, cnt = valid ? 1u : 0u, mx = fx, tail = (valid && pmax > 0.5f) ? 1u : 0u;        // <= 64 x 2^24: no overflow in 32 bits
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o, 64);
    cnt += __shfl_xor(cnt, o, 64);
    tail += __shfl_xor(tail, o, 64);
    const unsigned int other = __shfl_xor(mx, o, 64);
    mx = mx > other ? mx : other;
  }
  if ((threadIdx.x & 63) == 0 && cnt) {
    atomicAdd(stat + 4 * head, (unsigned long long)sum);
    atomicAdd(stat + 4 * head + 1, (unsigned long long)cnt);
    atomicMax(stat + 4 * head + 2, (unsigned long long)mx);
    if (tail) atomicAdd(stat + 4 * head + 3, (unsigned long long)tail);
  }
}

// The same statistic folded per WORKGROUP before it goes to memory (round 5: every batch is measured).  One set of device atomics per wave
// was 4096 atomics on sixteen addresses for a single-pair launch of the split-key kernel -- all arriving when the workgroups finish together:
// +14 us on a 61-us launch.  The waves add into a 32-byte LDS accumulator (`acc`: sum, count, max, tail; zeroed by peak_acc_init at kernel entry,
// ordered by the tile loop's barriers), peak_acc_flush sends one set per workgroup.  Integer sums: the totals are the same as before.
__device__ __forceinline__ void peak_acc_init(unsigned long long* acc) { if (threadIdx.x < 4) acc[threadIdx.x] = 0ull; }
__device__ __forceinline__ void peak_acc_add(unsigned long long* acc, float pmax, bool valid) {
  const unsigned int fx = valid ? (unsigned int)(fminf(fmaxf(pmax, 0.f), 1.f) * 16777216.f + 0.5f) : 0u;
  unsigned int sum = fx, cnt = valid ? 1u : 0u, mx = fx, tail = (valid && pmax > 0.5f) ? 1u : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o, 64);
    cnt += __shfl_xor(cnt, o, 64);
    tail += __shfl_xor(tail, o, 64);
    const unsigned int other = __shfl_xor(mx, o, 64);
    mx = mx > other ? mx : other;
  }
  if ((threadIdx.x & 63) == 0 && cnt) {
    atomicAdd(acc + 0, (unsigned long long)sum);
    atomicAdd(acc + 1, (unsigned long long)cnt);
    atomicMax(acc + 2, (unsigned long long)mx);
    if (tail) atomicAdd(acc + 3, (unsigned long long)tail);
  }
}
__device__ __forceinline__ void peak_acc_flush(unsigned long long* acc, unsigned long long* stat, int head) {      // (called by every live wave of the workgroup)
  __syncthreads();
  if (threadIdx.x == 0 && acc[1]) {
    atomicAdd(stat + 4 * head, acc[0]);
    atomicAdd(stat + 4 * head + 1, acc[1]);
    atomicMax(stat + 4 * head + 2, acc[2]);
    if (acc[3]) atomicAdd(stat + 4 * head + 3, acc[3]);
  }
}

template <int QP, bool F16 = false>   // 32-query blocks per wave: 1 (32 queries/wave, 128/workgroup) or 2 (64 / 256); F16: operands and P in IEEE half
__global__ __launch_bounds__(256) void attention_bf16_kernel(
    const uint16_t* __restrict__ qkv, int64_t ld, int q_col, int k_col, int v_col,
    const gims_attn_problem* __restrict__ problems, int n_groups, int n_heads, int n_qt, float* __restrict__ out,
    int64_t ld_out, uint16_t* __restrict__ out_hi, uint16_t* __restrict__ out_lo, int64_t ld_split, float c,
    unsigned long long* __restrict__ stat) {
  __shared__ __attribute__((aligned(16))) uint16_t Ks[2][KB * DH];      // double-buffered: one barrier per key tile
  __shared__ __attribute__((aligned(16))) uint16_t Vr[2][KB * VR_LD];      // row-major V tiles (LDS-DMA in, transposing reads out)
  __shared__ unsigned long long pacc[4];                                 // peakedness statistic of this workgroup (peak_acc_*)
  if (stat) peak_acc_init(pacc);
  constexpr int QWV = QW * QP;            // queries per wave
  constexpr int QBK = QWV * ATT_WAVES;    // queries per workgroup

  // XCD-aware order (workgroup b -> XCD b % 8, private L2 per XCD): all query tiles of one (problem, head) get
  // consecutive slots on ONE XCD, so its K/V panel is fetched from HBM once, not once per query tile.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int group = (slot / n_qt) * 8 + xcd;
  if (group >= n_groups) return;
  const gims_attn_problem pr = problems[group / n_heads];
  const int q0 = (slot % n_qt) * QBK;
  if (q0 >= pr.n_q) return;
  const int head = group % n_heads;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;

  // ---- Q^T B-operand fragments: lane holds Q[q = li][d = 16*s + 8*lh .. +8] for each of its QP query blocks
  bf16x8 qf[QP][4];
#pragma unroll
  for (int qi = 0; qi < QP; ++qi) {
    int qr = q0 + wave * QWV + qi * QW + li;
    qr = qr < pr.n_q ? qr : pr.n_q - 1;
    const uint16_t* qp = qkv + (int64_t)(pr.q_off + qr) * ld + q_col + head * DH + 8 * lh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[qi][s] = *(const bf16x8*)(qp + 16 * s);
  }

  f32x16 o[QP][2];  // O^T accumulators: d-block x 16 regs, column = query li
#pragma unroll
  for (int qi = 0; qi < QP; ++qi)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qi][i][r] = 0.f;
  // c = 1/sqrt(64) * log2(e), or 1 when the caller folded that factor into Q (GIMS_ATTN_Q_PRESCALED)
  // Deferred running max (rescale threshold): the accumulator is rescaled only when some row's tile max
  // exceeds its reference max by more than 2^DEFER in probability units; until then P is bounded by 2^DEFER
  // instead of 1, which bf16 (relative precision) and the f32 accumulators tolerate unchanged.
  constexpr float DEFER = 5.0f;
  const float defer_raw = DEFER / c;
  float m_run[QP], l_run[QP], m_true[QP];         // m_true: the row maximum itself (m_run lags it by up to DEFER), for `stat`
#pragma unroll
  for (int qi = 0; qi < QP; ++qi) { m_run[qi] = -1e30f; l_run[qi] = 0.f; m_true[qi] = -1e30f; }

  // staging: K tile 64 rows x 8 chunks = 512 chunks (2 per thread); V tile: key pair kp = t&31, d-octet t>>5
  // K goes global -> LDS by LDS-DMA: its staging registers, held across a whole tile, were being spilled to scratch (48 B per lane).  One DMA
  // instruction fills a contiguous KiB = 8 tile rows; the chunk swizzle of k_off is applied on the source side.  V^T still goes through registers.
  const int n_tiles = (pr.n_kv + KB - 1) / KB;
  const int wave_u = __builtin_amdgcn_readfirstlane(t >> 6);
  auto dma_k = [&](int kt, int buf) __attribute__((always_inline)) {
    const int kbase = kt * KB;
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      const int j = wave_u * 2 + j2, row = 8 * j + ((t & 63) >> 3), ch = (t & 7) ^ ((row >> 1) & 7);
      int kr = kbase + row; kr = kr < pr.n_kv ? kr : pr.n_kv - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qkv + (int64_t)(pr.kv_off + kr) * ld + k_col + head * DH + 8 * ch),
                                       (__attribute__((address_space(3))) void*)(Ks[buf] + j * 512), 16, 0, 0);
    }
  };
  // V the same way: row-major tiles with a 160-byte pitch; byte B = 1024 j + 16 lane of the padded image is (row B / 160, chunk (B % 160) / 16),
  // the two pad chunks of a row re-fetch chunk 0; ten instructions per tile (waves 0 and 1 issue three, waves 2 and 3 two)
  auto dma_v = [&](int kt, int buf) __attribute__((always_inline)) {
    const int kbase = kt * KB;
#pragma unroll
    for (int j3 = 0; j3 < 3; ++j3) {
      const int j = wave_u + 4 * j3;
      if (j < 10) {
        const int B = 1024 * j + 16 * (t & 63), row = B / (2 * VR_LD), pos = (B % (2 * VR_LD)) >> 4, ch = pos < 8 ? pos : 0;
        int kr = kbase + row; kr = kr < pr.n_kv ? kr : pr.n_kv - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qkv + (int64_t)(pr.kv_off + kr) * ld + v_col + head * DH + 8 * ch),
                                         (__attribute__((address_space(3))) void*)(Vr[buf] + j * 512), 16, 0, 0);
      }
    }
  };
  dma_k(0, 0);
  dma_v(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < n_tiles; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < n_tiles) { dma_k(kt + 1, (kt + 1) & 1); dma_v(kt + 1, (kt + 1) & 1); }

    // ---- S^T = K Q^T : two 32-key blocks x QP query blocks, K = 64 (4 steps of 16); K fragments shared by the query blocks
    f32x16 sacc[QP][2];
#pragma unroll
    for (int qi = 0; qi < QP; ++qi)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[qi][b][r] = 0.f;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = *(const bf16x8*)(Ks[buf] + k_off(b * 32 + li, 2 * s + lh));
#pragma unroll
        for (int qi = 0; qi < QP; ++qi) sacc[qi][b] = mfma_16b<F16>(kf, qf[qi][s], sacc[qi][b]);
      }
    __builtin_amdgcn_s_setprio(0);
    const int kbase = kt * KB;
    const bool partial = kbase + KB > pr.n_kv;
    bf16x8 pf[QP][4];
#pragma unroll
    for (int qi = 0; qi < QP; ++qi) {
      // ---- mask keys past the end (last tile only), tile max
      float tmax = -1e30f;
      if (partial) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kbase + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (key >= pr.n_kv) sacc[qi][b][r] = -1e30f;
          }
      }
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[qi][b][r]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      m_true[qi] = fmaxf(m_true[qi], tmax);
      if (__any(tmax > m_run[qi] + defer_raw)) {          // wave-uniform, rare after the first tiles
        const float m_new = fmaxf(m_run[qi], tmax);
        const float alpha = __builtin_amdgcn_exp2f((m_run[qi] - m_new) * c);
        m_run[qi] = m_new;
        l_run[qi] *= alpha;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[qi][i][r] *= alpha;
      }
      const float mc = m_run[qi] * c;
      // ---- P = exp2(S*c - m*c), packed to bf16 B-operand fragments in the lane's own key order
      float lsum = 0.f;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pv[r] = __builtin_amdgcn_exp2f(fmaf(sacc[qi][b][r], c, -mc));
          lsum += pv[r];
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {  // 16-key step h2 of block b: regs 8*h2 .. 8*h2+7
          uint4 pk;
          pk.x = pack_16b<F16>(pv[8 * h2 + 0], pv[8 * h2 + 1]);
          pk.y = pack_16b<F16>(pv[8 * h2 + 2], pv[8 * h2 + 3]);
          pk.z = pack_16b<F16>(pv[8 * h2 + 4], pv[8 * h2 + 5]);
          pk.w = pack_16b<F16>(pv[8 * h2 + 6], pv[8 * h2 + 7]);
          pf[qi][2 * b + h2] = __builtin_bit_cast(bf16x8, pk);
        }
      }
      l_run[qi] += lsum;
    }
    // ---- O^T += V^T P^T : two 32-d blocks, 4 steps of 16 keys; V^T fragments shared by the query blocks
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        // lane's k-slots of step s: keys 16s + {0-3, 8-11} + 4*lh
        const int voff = (4 * lh + ((lane & 15) >> 2)) * VR_LD + 16 * ((lane >> 4) & 1) + 4 * (lane & 3) + 16 * s * VR_LD + 32 * i;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Vr[buf] + voff));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Vr[buf] + voff + 8 * VR_LD));
        const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
        for (int qi = 0; qi < QP; ++qi) o[qi][i] = mfma_16b<F16>(vf, pf[qi][s], o[qi][i]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the DMA of the next K tile has landed
    __syncthreads();
  }

  // ---- normalise and store: lane holds query li, d = 32*i + 8*(r>>2) + 4*lh + (r&3)
#pragma unroll
  for (int qi = 0; qi < QP; ++qi) {
    const float l_tot = l_run[qi] + __shfl_xor(l_run[qi], 32, 64);
    const float inv = 1.f / l_tot;
    const int qr = q0 + wave * QWV + qi * QW + li;
    if (stat) peak_acc_add(pacc, __builtin_amdgcn_exp2f((m_true[qi] - m_run[qi]) * c) * inv, lh == 0 && qr < pr.n_q);
    if (qr < pr.n_q) {
      const int64_t grow = pr.q_off + qr;
      const int col0 = head * DH + 4 * lh;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = make_float4(o[qi][i][4 * g] * inv, o[qi][i][4 * g + 1] * inv, o[qi][i][4 * g + 2] * inv, o[qi][i][4 * g + 3] * inv);
          const int col = col0 + 32 * i + 8 * g;
          if (out) *(float4*)(out + grow * ld_out + col) = v;
          if (out_hi) {
            const uint32_t h01 = pack_bf2(v.x, v.y), h23 = pack_bf2(v.z, v.w);
            const uint32_t l01 = pack_bf2(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xffff0000u));
            const uint32_t l23 = pack_bf2(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xffff0000u));
            *(uint2*)(out_hi + grow * ld_split + spl_col(col)) = make_uint2(h01, h23);
            *(uint2*)(out_lo + grow * ld_split + spl_col(col)) = make_uint2(l01, l23);
          }
        }
    }
  }
  if (stat) peak_acc_flush(pacc, stat, head);
}


// ---------------------------------------------------------------------------------------------- 8-wave variant
// Same math and register layout as attention_bf16_kernel<2> (64 queries per wave), but a workgroup is EIGHT waves, two per
// SIMD (K/V tiles shared by 512 queries: half the staging traffic per query of the 4-wave kernel), and the softmax is
// stripped to what the measurements left standing:
//   * gfx950 does not overlap one wave's MFMAs with its SIMD-mate's VALU (tools/probes/phase_overlap_probe.hip), and a
//     wave issues one VALU instruction per ~4.9 cycles (valu_rate_probe.hip): a key tile costs its matrix cycles PLUS its
//     VALU issues.  Hence ONE instruction stream for all waves and one barrier per key tile (an earlier version kept the
//     two waves of a SIMD half a tile apart with four barriers -- 9 % slower), and as few VALU issues per score as possible:
//   * the softmax scale log2(e)/sqrt(64) is folded into Q by the caller (GIMS_ATTN_Q_PRESCALED, NOFMA) and the first pass is
//     "optimistic": P = exp2(S) with no running maximum at all (softmax is invariant to the reference point); row sums
//     outside [1e-30, 1e30] send the workgroup through a second, exact pass (running maximum with deferred rescale);
//   * row sums by v_pk_add_f32; V row-major in LDS, read transposed by ds_read_b64_tr_b16 (tr16_probe.hip);
//   * the split-bf16 output leaves as whole 256-byte rows through a wave-private LDS transpose.
constexpr int ATT8_WAVES = 8;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------- peakedness of a SAMPLE of rows
// The 8-wave kernel's optimistic softmax never looks at a maximum, so it cannot report the row maxima of P; making some of its workgroups
// take the exact pass instead cost 17 % of the launch (they become the tail of a two-round grid).  A SAMPLE is measured instead, by extra
// workgroups of the SAME launch (round 5: every batch is measured -- as a launch of its own the sample cost 16 us per layer at 2 x 4096 x 8, in
// the first workgroups of the attention grid it runs under the main workgroups): per (problem, head) group one workgroup takes 32 queries
// spread evenly over the group's queries and walks ALL keys -- wave w the 32-key tiles w, w + NW, ... -- with S^T = K Q^T on the matrix cores
// (K fragments straight from global memory: a lane's MFMA operand is one 16-byte piece of a key row), an exact online (maximum, sum) per
// query, and a merge of the partial pairs per query through LDS (xm, xl: [NW][64] floats each).  Row maximum of P = 1 / (sum of exp2(s - max)).
constexpr int PS_Q = 32;
template <bool F16, int NW>
__device__ __forceinline__ void attention_peak_sample_wg(const uint16_t* __restrict__ qkv, int64_t ld, int q_col, int k_col, const gims_attn_problem pr,
                                                         int head, float c, unsigned long long* __restrict__ stat, float* xm, float* xl) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int n_s = pr.n_q < PS_Q ? pr.n_q : PS_Q;                        // sampled queries: j -> row j * n_q / n_s
  if (n_s <= 0 || pr.n_kv <= 0) return;
  bf16x8 qf[4];
  {
    const int j = li < n_s ? li : n_s - 1;
    const int qr = (int)(((long long)j * pr.n_q) / n_s);
    const uint16_t* qp = qkv + (int64_t)(pr.q_off + qr) * ld + q_col + head * DH + 8 * lh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8*)(qp + 16 * s);
  }
  const int n_tiles = (pr.n_kv + 31) / 32;
  float m_run = -1e30f, l_run = 0.f;
  constexpr int TB = 4;                                                   // tiles in flight per wave (16 x 16-byte loads)
  for (int t0 = wave; t0 < n_tiles; t0 += NW * TB) {
    bf16x8 kf[TB][4];
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      const int t = t0 + b * NW;
      int kr = t * 32 + li;
      kr = kr < pr.n_kv ? kr : pr.n_kv - 1;
      const uint16_t* kp = qkv + (int64_t)(pr.kv_off + kr) * ld + k_col + head * DH + 8 * lh;
#pragma unroll
      for (int s = 0; s < 4; ++s) kf[b][s] = *(const bf16x8*)(kp + 16 * s);
    }
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      const int t = t0 + b * NW;
      if (t >= n_tiles) break;                                            // wave-uniform
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) sacc = mfma_16b<F16>(kf[b][s], qf[s], sacc);
      float tmax = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (key >= pr.n_kv) sacc[r] = -1e30f;
        tmax = fmaxf(tmax, sacc[r]);
      }
      const float m_new = fmaxf(m_run, tmax);
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f((sacc[r] - m_new) * c);
      l_run = l_run * __builtin_amdgcn_exp2f((m_run - m_new) * c) + sum;
      m_run = m_new;
    }
  }
  xm[wave * 64 + lane] = m_run;
  xl[wave * 64 + lane] = l_run;
  __syncthreads();
  if (wave == 0) {
    float m = -1e30f;
#pragma unroll
    for (int w = 0; w < NW; ++w) m = fmaxf(m, fmaxf(xm[w * 64 + li], xm[w * 64 + li + 32]));
    float l = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w)                                           // fixed order
      l += xl[w * 64 + li] * __builtin_amdgcn_exp2f((xm[w * 64 + li] - m) * c) + xl[w * 64 + li + 32] * __builtin_amdgcn_exp2f((xm[w * 64 + li + 32] - m) * c);
    emit_peak_stat(stat, head, 1.f / l, lh == 0 && li < n_s);
  }
}

template <bool PROF, bool NOFMA, bool F16 = false>     // NOFMA: Q carries the softmax scale (c == 1): the optimistic pass is P = exp2(S), reference 0
__global__ __launch_bounds__(512) void attention8_bf16_kernel(
    const uint16_t* __restrict__ qkv, int64_t ld, int q_col, int k_col, int v_col,
    const gims_attn_problem* __restrict__ problems, int n_groups, int n_heads, int n_qt, float* __restrict__ out,
    int64_t ld_out, uint16_t* __restrict__ out_hi, uint16_t* __restrict__ out_lo, int64_t ld_split, unsigned long long* prof,
    int exact_only, float c, unsigned long long* __restrict__ stat, int n_sample) {
  // diagnostics (GIMS_ATTN_PROF=1): cycles of wave 0 (group A) and wave 4 (group B) of workgroup 0 per phase, split into
  // work (phase start -> barrier reached) and wait (inside the barrier)
  unsigned long long pt = 0, pacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_entry = PROF ? __builtin_readcyclecounter() : 0, r_entry = PROF ? __builtin_amdgcn_s_memrealtime() : 0;   // whole-kernel span of this wave
  unsigned long long t_loop0 = 0, t_loop1 = 0;
  auto stamp_work = [&](int ph) __attribute__((always_inline)) { if (PROF) { const unsigned long long n = __builtin_readcyclecounter(); pacc[2 * ph] += n - pt; pt = n; } };
  // sub-stamps of the staging phase: [8] = its matrix segment, [9] = store_tile (the rest of the phase's work is load_tile)
  auto stamp_sub = [&](int i) __attribute__((always_inline)) {
    if (PROF) { __builtin_amdgcn_sched_barrier(0); const unsigned long long n = __builtin_readcyclecounter(); pacc[i] += n - pt; pt = n; __builtin_amdgcn_sched_barrier(0); } };
  auto stamp_wait = [&](int ph) __attribute__((always_inline)) { if (PROF) { const unsigned long long n = __builtin_readcyclecounter(); pacc[2 * ph + 1] += n - pt; pt = n; } };
  // Barriers inside the tile loop are RAW s_barrier + lgkmcnt(0): __syncthreads() also waits vmcnt(0), which would stall
  // every wave at the first barrier after load_tile until the global loads of the NEXT tile have landed.  The loads are
  // only needed by store_tile three phases later (register dependency).
  auto raw_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
#define ATT8_BAR(ph) do { stamp_work(ph); raw_barrier(); stamp_wait(ph); } while (0)
  __shared__ __attribute__((aligned(16))) uint16_t Ks[2][KB * DH];
  // V tile row-major [key][d] with a 192-byte pitch (conflict-free for the transposing reads of seg_pv, see there)
  constexpr int VROW = 96;
  __shared__ __attribute__((aligned(16))) uint16_t Vs[2][KB * VROW];
  constexpr int QP = 2, QWV = QW * QP, QBK = QWV * ATT8_WAVES;   // 64 queries per wave, 512 per workgroup

  // measured launches (stat != NULL): the first n_sample workgroups (a multiple of 8: the XCD mapping of the others is unchanged) measure the
  // peakedness of a sample of rows, one (problem, head) group each, and are gone long before the main workgroups finish
  if (stat != nullptr && (int)blockIdx.x < n_sample) {
    if ((int)blockIdx.x < n_groups)
      attention_peak_sample_wg<F16, ATT8_WAVES>(qkv, ld, q_col, k_col, problems[blockIdx.x / n_heads], (int)blockIdx.x % n_heads, c, stat, (float*)&Ks[0][0],
                                                (float*)&Ks[1][0]);
    return;
  }
  const int bid = (int)blockIdx.x - (stat != nullptr ? n_sample : 0);
  const int xcd = bid & 7, slot = bid >> 3;
  const int group = (slot / n_qt) * 8 + xcd;
  if (group >= n_groups) return;
  const gims_attn_problem pr = problems[group / n_heads];
  const int q0 = (slot % n_qt) * QBK;
  if (q0 >= pr.n_q) return;
  const int head = group % n_heads;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 31, lh = lane >> 5;

  // staging: K and V tiles are 64 rows x 8 chunks of 16 bytes = 512 chunks each, one of each per thread, same (row, chunk):
  // eight lanes cover one 128-byte line of K and one of V, and the two addresses differ by a constant
  uint4 rk, rv;
  const int n_tiles = (pr.n_kv + KB - 1) / KB;
  auto load_tile = [&](int kt) __attribute__((always_inline)) {
    const int kbase = kt * KB;
    const int row = t >> 3, ch = t & 7;
    int kr = kbase + row; kr = kr < pr.n_kv ? kr : pr.n_kv - 1;
    // wave-uniform 64-bit base (the problem's first key row) + a 32-bit element offset per lane: the loads take the scalar-base addressing
    // form and no 64-bit per-lane address stays alive across the tile (n_kv * ld < 2^32: ld <= 16384 is checked by the launcher)
    const uint16_t* base = qkv + (int64_t)pr.kv_off * ld + head * DH;
    const uint32_t off = (uint32_t)kr * (uint32_t)ld + 8u * (uint32_t)ch;
    rk = *(const uint4*)(base + k_col + off);
    rv = *(const uint4*)(base + v_col + off);
  };
  load_tile(0);                                // in flight together with the Q fragments below
  bool tile0_requested = true;
  // The epilogue's transposing slices (Es) and, in the half instance, the workgroup's Q rows (Qs) share one LDS array: Q is dead when the
  // epilogue starts (a workgroup barrier separates them).  Half instance: the 32 registers of the Q fragments are needed for the reference
  // blocks `nref` below, so Q lives in LDS (each wave its own 64 rows, 128 bytes each, 16-byte chunks swizzled like the K tile) and its
  // fragments are re-read per key tile together with the K fragments.
  __shared__ __attribute__((aligned(16))) uint16_t QE[ATT8_WAVES * 32 * 132];
  static_assert(ATT8_WAVES * 32 * 132 >= 512 * DH, "the Q rows of a workgroup fit the epilogue staging array");
  bf16x8 qf[QP][4];
#pragma unroll
  for (int qi = 0; qi < QP; ++qi) {
    int qr = q0 + wave * QWV + qi * QW + li;
    qr = qr < pr.n_q ? qr : pr.n_q - 1;
    const uint16_t* qp = qkv + (int64_t)(pr.q_off + qr) * ld + q_col + head * DH + 8 * lh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[qi][s] = *(const bf16x8*)(qp + 16 * s);
  }
  if constexpr (F16) {
#pragma unroll
    for (int qi = 0; qi < QP; ++qi)
#pragma unroll
      for (int s = 0; s < 4; ++s) *(bf16x8*)(QE + (wave * QWV) * DH + k_off(qi * QW + li, 2 * s + lh)) = qf[qi][s];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // wave-private rows: no workgroup barrier
  }
  // Q must have LANDED before the tile loop: otherwise the compiler guards the first MFMAs of EVERY iteration with
  // s_waitcnt vmcnt(..0), and since vmcnt retires in order those waits also sit out the loads of the next K/V tile that
  // were issued a moment earlier (measured: ~1100 of the 1575 cycles of that phase)
  __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0) only
  f32x16 o[QP][2];
  // half instance: -reference of each query block, sixteen equal registers per block -- the C operand of the first MFMA of every S^T chain, so
  // the scores arrive as S - ref and the exponentials need no subtraction (zero in the exact pass, whose softmax carries its own reference)
  f32x16 nref[F16 ? QP : 1];
  constexpr float DEFER = 5.0f;
  const float defer_raw = DEFER / c;
  float m_run[QP], l_run[QP];
  // bf16 instance, measured launches: the largest share of a row's mass that one 32-key half tile (this lane's keys of a tile) has held -- an upper
  // bound of the row's largest probability at ONE v_max per query block and tile, so that a single sharply peaked row inside a diffuse layer
  // cannot hide behind the 32-query sample (the guard of attention_precision='auto' looks at the head's largest row maximum)
  float hmass[QP];

  auto store_tile = [&](int buf) __attribute__((always_inline)) {
    *(uint4*)(Ks[buf] + k_off(t >> 3, t & 7)) = rk;
    *(uint4*)(Vs[buf] + (t >> 3) * VROW + 8 * (t & 7)) = rv;
  };

  f32x16 sacc[QP][2];
  bf16x8 pf[QP][4];
  auto seg_qk = [&](int kt) __attribute__((always_inline)) {                 // S^T = K Q^T of tile kt (16 MFMAs)
    const int buf = kt & 1;
    if constexpr (F16) {
      // K and Q fragments from LDS (sixteen reads in flight), every chain seeded with -ref: S^T - ref
      bf16x8 kf[2][4], qh[QP][4];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[b][s] = *(const bf16x8*)(Ks[buf] + k_off(b * 32 + li, 2 * s + lh));
#pragma unroll
      for (int qi = 0; qi < QP; ++qi)
#pragma unroll
        for (int s = 0; s < 4; ++s) qh[qi][s] = *(const bf16x8*)(QE + (wave * QWV) * DH + k_off(qi * QW + li, 2 * s + lh));
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int qi = 0; qi < QP; ++qi) sacc[qi][b] = mfma_16b<F16>(kf[b][s], qh[qi][s], s == 0 ? nref[qi] : sacc[qi][b]);
      // (left to the compiler's order -- four reads, two MFMAs, ...: pinning all sixteen reads ahead of the MFMAs like the bf16 form below measured
      // 324 us per launch against 316)
      return;
    }
#pragma unroll
    for (int qi = 0; qi < QP; ++qi)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[qi][b][r] = 0.f;
    // all 8 K fragments are requested before the first MFMA (P is dead here, its 32 registers are free): one LDS latency
    // per segment instead of one per fragment -- only ONE wave per SIMD is in a matrix segment, nobody else hides it
    bf16x8 kf[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int s = 0; s < 4; ++s) kf[b][s] = *(const bf16x8*)(Ks[buf] + k_off(b * 32 + li, 2 * s + lh));
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int qi = 0; qi < QP; ++qi) sacc[qi][b] = mfma_16b<F16>(kf[b][s], qf[qi][s], sacc[qi][b]);
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);     // 8 LDS reads ...
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);    // ... then the 16 MFMAs (the scheduler otherwise serialises read -> wait -> MFMA)
  };
  auto seg_softmax = [&](int kt, int qi, bool exact) __attribute__((always_inline)) {     // online softmax of query block qi on tile kt
    const int kbase = kt * KB;
    if (kbase + KB > pr.n_kv) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key >= pr.n_kv) sacc[qi][b][r] = -1e30f;
        }
    }
    // Reference point of the exponentials.  softmax is invariant to it, so the OPTIMISTIC pass takes the row maximum of
    // the first key tile and never looks at a maximum again (32 v_max + a cross-lane exchange + the rescale test per
    // block and tile are a fifth of the VALU work that bounds this kernel): later scores above the reference just give
    // p > 1.  Only a score more than ~100 octaves above it could overflow; the row sums are checked after the loop and the
    // workgroup then repeats its tiles in the exact mode (running maximum, deferred rescale).
    // With the scale folded into Q (NOFMA) the optimistic pass needs no reference at all: P = exp2(S) as the MFMA left it
    // (row sums outside [1e-30, 1e30] send the workgroup to the exact pass) -- 64 fewer VALU per wave and tile again.
    if constexpr (F16) {
      if (!exact) {
        // IEEE half has range 65504, so P = exp2(S) cannot go unreferenced like the bf16 pass below.  A row reference follows the running
        // maximum with a deferred rescale (only when a tile's maximum exceeds the reference by more than 10 octaves: P stays below 2^10),
        // and it costs NO instruction per score: it enters through the accumulator seed of the S^T chains (seg_qk), so the scores arrive
        // as S - ref and P = exp2 of them as they are.  The only branch encloses the (rare) raise -- rescale of O and l, shift of this
        // tile's scores, new seed -- NOT the exponentials: those stay in one basic block with the P V MFMAs that follow, like the bf16
        // pass (the compiler interleaves them).  The first tile always sets the reference to its row maxima (they may lie far below the
        // starting reference 0, where exp2(S) would flush to zero).  The reference never exceeds the true row maximum, so the dominant
        // keys sit in half's normal range (11-bit significand); keys 2^-14 below the reference go subnormal at an absolute 2^-25.
        // Cost against the reference-free bf16 pass (GIMS_ATTN_PROF=1, wave 4 of workgroup 0, 64 tiles): S^T segment 68 k cycles against 53 k (the
        // Q fragments come from LDS: their registers hold the seeds), the two softmax segments + P V 177 k against 141 k (sixteen v_max3 and
        // one wave-uniform test per query block and tile) -- 316 us per launch against 265.  (Measured out in round 4, all at 315 +- 2 us as
        // well: the reference as a per-score subtraction (v_fma_f32, then v_pk_add_f32) with Q in registers and a lazy raise detected from
        // the row sums, the exponentials inside a retry loop; and, 400 us, that retry as an inlined cold path: 96-220 bytes of scratch per
        // lane whose reloads put a vmcnt(0) behind the next tile's global loads.)
        constexpr float RAISE = 10.f;
        float tmax = -1e30f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[qi][b][r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        if (kt == 0 || __any(tmax * c > RAISE)) {
          const float dmax = kt == 0 ? tmax : fmaxf(tmax, 0.f);                 // (after the first tile the reference only ever moves up)
          // (first tile: the sums and the output are still zero -- nothing to rescale; exp2 of a first-tile maximum below -128 octaves would be
          // +inf, 0 * inf = NaN in l_run and o, and the whole workgroup would repeat in the exact pass: a hidden 2 x on large-magnitude logits)
          const float alpha = kt == 0 ? 1.f : __builtin_amdgcn_exp2f(-dmax * c);
          m_run[qi] += dmax;
          l_run[qi] *= alpha;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qi][i][r] *= alpha;
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[qi][b][r] -= dmax;
#pragma unroll
          for (int r = 0; r < 16; ++r) nref[qi][r] = -m_run[qi];
        }
        f32x2 lsum2 = {0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float pv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) pv[r] = NOFMA ? __builtin_amdgcn_exp2f(sacc[qi][b][r]) : __builtin_amdgcn_exp2f(sacc[qi][b][r] * c);
#pragma unroll
          for (int r = 0; r < 16; r += 2) lsum2 += f32x2{pv[r], pv[r + 1]};
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            uint4 pk;
            pk.x = pack_16b<F16>(pv[8 * h2 + 0], pv[8 * h2 + 1]);
            pk.y = pack_16b<F16>(pv[8 * h2 + 2], pv[8 * h2 + 3]);
            pk.z = pack_16b<F16>(pv[8 * h2 + 4], pv[8 * h2 + 5]);
            pk.w = pack_16b<F16>(pv[8 * h2 + 6], pv[8 * h2 + 7]);
            pf[qi][2 * b + h2] = __builtin_bit_cast(bf16x8, pk);
          }
        }
        l_run[qi] += lsum2.x + lsum2.y;
        return;
      }
    }
    const bool track = exact || (!NOFMA && kt == 0);
    if (track) {
      float tmax = -1e30f;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[qi][b][r]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      if (__any(tmax > m_run[qi] + defer_raw)) {          // wave-uniform, rare after the first tiles
        const float m_new = fmaxf(m_run[qi], tmax);
        const float alpha = __builtin_amdgcn_exp2f((m_run[qi] - m_new) * c);
        m_run[qi] = m_new;
        l_run[qi] *= alpha;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[qi][i][r] *= alpha;
      }
    }
    const float mc = m_run[qi] * c;
    f32x2 lsum2 = {0.f, 0.f};                   // packed adds: one VALU issue per score pair
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        pv[r] = (NOFMA && !exact) ? __builtin_amdgcn_exp2f(sacc[qi][b][r]) : __builtin_amdgcn_exp2f(fmaf(sacc[qi][b][r], c, -mc));
#pragma unroll
      for (int r = 0; r < 16; r += 2) lsum2 += f32x2{pv[r], pv[r + 1]};
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        uint4 pk;
        pk.x = pack_16b<F16>(pv[8 * h2 + 0], pv[8 * h2 + 1]);
        pk.y = pack_16b<F16>(pv[8 * h2 + 2], pv[8 * h2 + 3]);
        pk.z = pack_16b<F16>(pv[8 * h2 + 4], pv[8 * h2 + 5]);
        pk.w = pack_16b<F16>(pv[8 * h2 + 6], pv[8 * h2 + 7]);
        pf[qi][2 * b + h2] = __builtin_bit_cast(bf16x8, pk);
      }
    }
    const float lsum = lsum2.x + lsum2.y;
    l_run[qi] += lsum;
    if (!F16 && !exact) hmass[qi] = fmaxf(hmass[qi], lsum);
  };
  auto seg_pv = [&](int kt) __attribute__((always_inline)) {                  // O^T += V^T P^T of tile kt (16 MFMAs)
    const int buf = kt & 1;
    // V^T fragments straight from the row-major tile with the transposing LDS read (ds_read_b64_tr_b16): within a
    // 16-lane group, lane i receives as element j the element (i & 3) of the 8-byte chunk addressed by lane 4j + (i >> 2)
    // (tools/probes/tr16_probe.hip).  A group addresses the block [4 keys][16 d] -- lane i: key i >> 2, d 4 (i & 3).. --
    // and gets back column d = i with the 4 keys; MFMA s wants, in k-slots j = 0..7 of lane (d = li, lh), the keys
    // 16 s + 8 (j >> 2) + 4 lh + (j & 3) (the order P already has): two such reads 8 keys apart.  One base address per
    // lane, everything else is an immediate offset.  Pitch 192 B: the four key rows of a read land on disjoint quarters
    // of the 64 banks.
    bf16x8 vf[4][2];                           // S is dead here: all 8 V fragments in flight before the first MFMA
    const uint16_t* vb = Vs[buf] + (4 * lh + ((lane & 15) >> 2)) * VROW + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vb + 16 * s * VROW + 32 * i));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vb + (16 * s + 8) * VROW + 32 * i));
        vf[s][i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int qi = 0; qi < QP; ++qi) o[qi][i] = mfma_16b<F16>(vf[s][i], pf[qi][s], o[qi][i]);
    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);    // the V fragments are two 8-byte transposing reads each
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
  };

  // one optimistic pass; a second, exact one only if a row sum overflowed (see seg_softmax).  Two instantiations of the
  // tile loop rather than a loop around it: the restart edge cost 34 spilled registers.
  auto pass = [&](auto ex) __attribute__((always_inline)) -> bool {
  constexpr bool exact = decltype(ex)::value;
#pragma unroll
  for (int qi = 0; qi < QP; ++qi) {
    m_run[qi] = (NOFMA && !exact && !F16) ? 0.f : -1e30f; l_run[qi] = 0.f; hmass[qi] = 0.f;
    if constexpr (F16) {
      // first pass: start from reference 0 (a seed of 1e30 would absorb S); the first tile's softmax sets it to the tile's row maxima
#pragma unroll
      for (int r = 0; r < 16; ++r) nref[qi][r] = 0.f;
      if (!exact) m_run[qi] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qi][i][r] = 0.f;
  }
  if (!tile0_requested) load_tile(0);           // (second, exact pass)
  tile0_requested = false;
  store_tile(0);
  if (1 < n_tiles) load_tile(1);
  __syncthreads();
  // One stream for all eight waves, ONE barrier per key tile (the double-buffered tile t+1 is stored after this wave's last
  // read of tile t-1 -- the previous barrier -- and read after the next one).  The two waves of a SIMD drift freely inside a
  // tile; the earlier design kept them half a tile apart with four barriers ("A multiplies while B exponentiates"), which
  // the overlap probes showed to buy nothing (section 4.2 of DESIGN.md) and this order beats by 9 %.
  if (PROF) { pt = __builtin_readcyclecounter(); t_loop0 = pt; }
  for (int kt = 0; kt < n_tiles; ++kt) {
    seg_qk(kt);
    stamp_work(0);
    seg_softmax(kt, 0, exact);
    stamp_work(1);
    seg_softmax(kt, 1, exact);
    stamp_work(2);
    seg_pv(kt);
    stamp_sub(8);
    if (kt + 1 < n_tiles) store_tile((kt + 1) & 1);
    stamp_sub(9);
    if (kt + 2 < n_tiles) load_tile(kt + 2);
    ATT8_BAR(3);
  }
  if (PROF) t_loop1 = __builtin_readcyclecounter();
  bool bad = false;
#pragma unroll
  for (int qi = 0; qi < QP; ++qi) bad = bad || !(l_run[qi] < 1e30f) || (NOFMA && !(l_run[qi] > 1e-30f));      // inf / NaN / implausible
  return bad;
  };
  // (__syncthreads_or is also the barrier that lets the second pass overwrite the last tiles in LDS)
  bool ran_exact = (exact_only & 1) != 0;
  if (!ran_exact) ran_exact = __syncthreads_or(pass(std::false_type{})) != 0;
  if (ran_exact) pass(std::true_type{});
  if constexpr (!F16) {
    if (stat != nullptr && (ran_exact || pr.n_kv >= 512)) {
      // rows whose bound reaches 1/2 (none in a diffuse layer: the block below is skipped by the whole wave); a workgroup that needed the exact
      // pass -- a row sum left f32's range: scores ~100 octaves apart -- reports 1
      unsigned fx = 0u;
#pragma unroll
      for (int qi = 0; qi < QP; ++qi) {
        const float l_tot = l_run[qi] + __shfl_xor(l_run[qi], 32, 64);
        const float h = fmaxf(hmass[qi], __shfl_xor(hmass[qi], 32, 64));
        const bool row = q0 + wave * QWV + qi * QW + li < pr.n_q;
        const float frac = ran_exact ? 1.f : fminf(h / l_tot, 1.f);
        if (row && frac >= 0.5f) { const unsigned f = (unsigned)(frac * 16777216.f + 0.5f); fx = fx > f ? fx : f; }
      }
      if (__ballot(fx != 0u) != 0ull) {
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) { const unsigned other = __shfl_xor(fx, o2, 64); fx = fx > other ? fx : other; }
        if (lane == 0) atomicMax(stat + 4 * head + 2, (unsigned long long)fx);
      }
    }
  }
  if (PROF && bid == 0 && (wave == 0 || wave == 4) && lane == 0)
    for (int i = 0; i < 10; ++i) prof[(wave >> 2) * 10 + i] = pacc[i];

  // ---- normalise and store.  In the MFMA layout a lane owns 4 consecutive channels of ONE query, so direct stores scatter
  // 8-byte pieces over 32 rows per instruction (store-issue-bound: ~10k cycles per workgroup, 15 % of the kernel at 1024
  // keys).  For the split-bf16 output -- a head's 64 channels are 256 contiguous bytes of a row there, [32 hi|32 lo] x 2 --
  // each wave transposes one 32-query block at a time through a private LDS slice (264-byte pitch: the 8-byte writes of 16
  // lanes hit 32 distinct banks) and stores whole rows, 16 bytes per lane, 4 rows per instruction.
  const bool row_stores = out_hi && !out && out_lo == out_hi + 32;
#pragma unroll
  for (int qi = 0; qi < QP; ++qi) {
    const float l_tot = l_run[qi] + __shfl_xor(l_run[qi], 32, 64);
    const float inv = 1.f / l_tot;
    const int qr = q0 + wave * QWV + qi * QW + li;
    if (row_stores) {
      uint16_t* es = QE + wave * (32 * 132);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = make_float4(o[qi][i][4 * g] * inv, o[qi][i][4 * g + 1] * inv, o[qi][i][4 * g + 2] * inv, o[qi][i][4 * g + 3] * inv);
          const uint32_t h01 = pack_bf2(v.x, v.y), h23 = pack_bf2(v.z, v.w);
          const uint32_t l01 = pack_bf2(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xffff0000u));
          const uint32_t l23 = pack_bf2(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xffff0000u));
          uint16_t* e = es + li * 132 + 64 * i + 8 * g + 4 * lh;          // block i of the row: hi at +0, lo at +32
          *(uint2*)e = make_uint2(h01, h23);
          *(uint2*)(e + 32) = make_uint2(l01, l23);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // wave-private slice: no workgroup barrier
      const int c = lane & 15, rs = lane >> 4;                            // 16-byte chunk of the row, row within the group of 4
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = 4 * it + rs;
        const uint4 w = *(const uint4*)(es + r * 132 + 8 * c);
        const int qrow = q0 + wave * QWV + qi * QW + r;
        if (qrow < pr.n_q) *(uint4*)(out_hi + (int64_t)(pr.q_off + qrow) * ld_split + spl_col(head * DH) + 8 * c) = w;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // read-back done before the next block overwrites
    } else if (qr < pr.n_q) {
      const int64_t grow = pr.q_off + qr;
      const int col0 = head * DH + 4 * lh;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = make_float4(o[qi][i][4 * g] * inv, o[qi][i][4 * g + 1] * inv, o[qi][i][4 * g + 2] * inv, o[qi][i][4 * g + 3] * inv);
          const int col = col0 + 32 * i + 8 * g;
          if (out) *(float4*)(out + grow * ld_out + col) = v;
          if (out_hi) {
            const uint32_t h01 = pack_bf2(v.x, v.y), h23 = pack_bf2(v.z, v.w);
            const uint32_t l01 = pack_bf2(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xffff0000u));
            const uint32_t l23 = pack_bf2(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xffff0000u));
            *(uint2*)(out_hi + grow * ld_split + spl_col(col)) = make_uint2(h01, h23);
            *(uint2*)(out_lo + grow * ld_split + spl_col(col)) = make_uint2(l01, l23);
          }
        }
    }
  }
  if (PROF && bid == 0 && wave == 0 && lane == 0) {      // spans of wave 0: prologue, tile loop, epilogue (shader cycles) and the whole kernel on the 100-MHz counter
    const unsigned long long t_exit = __builtin_readcyclecounter(), r_exit = __builtin_amdgcn_s_memrealtime();
    prof[20] = t_loop0 - t_entry; prof[21] = t_loop1 - t_loop0; prof[22] = t_exit - t_loop1; prof[23] = r_exit - r_entry;
  }
}

// ---------------------------------------------------------------------------------------------- split-bf16 (x3) variant
// GIMS_ATTN_X3: the same flash attention at f32-class accuracy for SHARPLY PEAKED softmaxes (trained weights; the
// 'peaked' reference goldens): with plain bf16 operands a logit of magnitude ~20 carries an absolute error of ~0.03
// (2^-9 relative), i.e. a few per cent on the probabilities, and a softmax dominated by one key passes V's and P's own
// bf16 roundings (2^-9) straight into the message.  Here every operand is a split-bf16 pair (x = hi + lo, 2^-17
// relative) and every product is three MFMAs, like the linear layers:
//     S^T  = Kh Qh^T + Kh Ql^T + Kl Qh^T            O^T += Vh^T Ph^T + Vh^T Pl^T + Vl^T Ph^T
// Q, K, V are read from the SPL32 split buffer the 3-pass Q/K/V GEMM writes ([rows][pitch >= 2*768]: per row and
// 32-channel block 32 hi then 32 lo values); P is split in registers.  Geometry of attention_bf16_kernel<1> (4 waves x
// 32 queries, 64-key tiles, running maximum with deferred rescale); LDS is dynamic (67 KB: hi and lo planes of the
// double-buffered K and V^T tiles).
constexpr int X3W_VR = VR_LD;              // row-major V tiles of the split-bf16 kernels
constexpr int X3_LDS_BYTES = 2 * 2 * (KB * DH + KB * X3W_VR) * 2;      // 72 KB: hi and lo planes of the double-buffered K and V tiles

__device__ __forceinline__ void attention_x3_block(
    const uint16_t* __restrict__ qkv, int64_t ld, int q_col, int k_col, int v_col,
    const gims_attn_problem* __restrict__ problems, int n_groups, int n_heads, int n_qt, float* __restrict__ out,
    int64_t ld_out, uint16_t* __restrict__ out_hi, uint16_t* __restrict__ out_lo, int64_t ld_split, float c,
    unsigned long long* __restrict__ stat, const int bid) {
  extern __shared__ __attribute__((aligned(16))) uint16_t x3_lds[];
  // [plane p = hi/lo][buffer]: K tiles then V^T tiles
  auto Ks = [&](int p, int buf) __attribute__((always_inline)) { return x3_lds + (p * 2 + buf) * (KB * DH); };
  // V tiles row-major [key][d], 160-byte pitch, filled by LDS-DMA and read transposed (see attention_x3w_kernel)
  auto Vr = [&](int p, int buf) __attribute__((always_inline)) { return x3_lds + 4 * (KB * DH) + (p * 2 + buf) * (KB * X3W_VR); };

  const int xcd = bid & 7, slot = bid >> 3;
  const int group = (slot / n_qt) * 8 + xcd;
  if (group >= n_groups) return;
  const gims_attn_problem pr = problems[group / n_heads];
  const int q0 = (slot % n_qt) * QB;
  if (q0 >= pr.n_q) return;
  const int head = group % n_heads;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  // a head's 64 channels are two 32-channel SPL32 blocks: channel d of the head sits at spl_col(col + head*64 + d) (hi), +32 (lo)
  const int qs = spl_col(q_col + head * DH), ks = spl_col(k_col + head * DH), vs = spl_col(v_col + head * DH);
  auto doff = [&](int d8) __attribute__((always_inline)) { return ((d8 >> 2) << 6) + ((d8 & 3) << 3); };   // 8-channel chunk d8 of the head

  bf16x8 qh[4], ql[4];
  {
    int qr = q0 + wave * QW + li;
    qr = qr < pr.n_q ? qr : pr.n_q - 1;
    const uint16_t* qp = qkv + (int64_t)(pr.q_off + qr) * ld + qs;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qh[s] = *(const bf16x8*)(qp + doff(2 * s + lh));
      ql[s] = *(const bf16x8*)(qp + doff(2 * s + lh) + 32);
    }
  }
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  constexpr float DEFER = 5.0f;
  const float defer_raw = DEFER / c;
  float m_run = -1e30f, l_run = 0.f, m_true = -1e30f;

  const int n_tiles = (pr.n_kv + KB - 1) / KB;
  // K goes global -> LDS by LDS-DMA, the chunk swizzle of k_off applied on the source side (see attention_x3w_kernel); V^T through registers
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto dma_k = [&](int kt, int buf) __attribute__((always_inline)) {
    const int kbase = kt * KB;
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      const int j = wave_u * 2 + j2, row = 8 * j + (lane >> 3), ch = (lane & 7) ^ ((row >> 1) & 7);
      int kr = kbase + row; kr = kr < pr.n_kv ? kr : pr.n_kv - 1;
      const uint16_t* src = qkv + (int64_t)(pr.kv_off + kr) * ld + ks + doff(ch);
#pragma unroll
      for (int p = 0; p < 2; ++p)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 32 * p),
                                         (__attribute__((address_space(3))) void*)(Ks(p, buf) + j * 512), 16, 0, 0);
    }
  };
  auto dma_v = [&](int kt, int buf) __attribute__((always_inline)) {
    const int kbase = kt * KB;
#pragma unroll
    for (int j3 = 0; j3 < 3; ++j3) {
      const int j = wave_u + 4 * j3;
      if (j < 10) {
        const int B = 1024 * j + 16 * lane, row = B / (2 * X3W_VR), pos = (B % (2 * X3W_VR)) >> 4, ch = pos < 8 ? pos : 0;
        int kr = kbase + row; kr = kr < pr.n_kv ? kr : pr.n_kv - 1;
        const uint16_t* src = qkv + (int64_t)(pr.kv_off + kr) * ld + vs + doff(ch);
#pragma unroll
        for (int p = 0; p < 2; ++p)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 32 * p),
                                           (__attribute__((address_space(3))) void*)(Vr(p, buf) + j * 512), 16, 0, 0);
      }
    }
  };
  dma_k(0, 0);
  dma_v(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < n_tiles; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < n_tiles) { dma_k(kt + 1, buf ^ 1); dma_v(kt + 1, buf ^ 1); }
    f32x16 sacc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[b][r] = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kh = *(const bf16x8*)(Ks(0, buf) + k_off(b * 32 + li, 2 * s + lh));
        const bf16x8 kl = *(const bf16x8*)(Ks(1, buf) + k_off(b * 32 + li, 2 * s + lh));
        sacc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[s], sacc[b], 0, 0, 0);      // small terms first
        sacc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[s], sacc[b], 0, 0, 0);
        sacc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[s], sacc[b], 0, 0, 0);
      }
    const int kbase = kt * KB;
    if (kbase + KB > pr.n_kv) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key >= pr.n_kv) sacc[b][r] = -1e30f;
        }
    }
    float tmax = -1e30f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[b][r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    m_true = fmaxf(m_true, tmax);
    if (__any(tmax > m_run + defer_raw)) {
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    }
    const float mc = m_run * c;
    bf16x8 ph[4], pl[4];
    float lsum = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pv[r] = __builtin_amdgcn_exp2f(fmaf(sacc[b][r], c, -mc));
        lsum += pv[r];
      }
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        uint32_t hw[4], lw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x = pv[8 * h2 + 2 * j], y = pv[8 * h2 + 2 * j + 1];
          hw[j] = pack_bf2(x, y);
          lw[j] = pack_bf2(x - __uint_as_float(hw[j] << 16), y - __uint_as_float(hw[j] & 0xffff0000u));
        }
        ph[2 * b + h2] = __builtin_bit_cast(bf16x8, make_uint4(hw[0], hw[1], hw[2], hw[3]));
        pl[2 * b + h2] = __builtin_bit_cast(bf16x8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
      }
    }
    l_run += lsum;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int voff = (4 * lh + ((lane & 15) >> 2)) * X3W_VR + 16 * ((lane >> 4) & 1) + 4 * (lane & 3) + 16 * s * X3W_VR + 32 * i;
        const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Vr(0, buf) + voff));
        const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Vr(0, buf) + voff + 8 * X3W_VR));
        const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Vr(1, buf) + voff));
        const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Vr(1, buf) + voff + 8 * X3W_VR));
        const bf16x8 vh = __builtin_bit_cast(bf16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
        const bf16x8 vl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
        o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph[s], o[i], 0, 0, 0);
        o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl[s], o[i], 0, 0, 0);
        o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph[s], o[i], 0, 0, 0);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the DMA of the next K tile has landed
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  const int qr = q0 + wave * QW + li;
  if (stat) emit_peak_stat(stat, head, __builtin_amdgcn_exp2f((m_true - m_run) * c) * inv, lh == 0 && qr < pr.n_q);
  if (qr < pr.n_q) {
    const int64_t grow = pr.q_off + qr;
    const int col0 = head * DH + 4 * lh;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = make_float4(o[i][4 * g] * inv, o[i][4 * g + 1] * inv, o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv);
        const int col = col0 + 32 * i + 8 * g;
        if (out) *(float4*)(out + grow * ld_out + col) = v;
        if (out_hi) {
          const uint32_t h01 = pack_bf2(v.x, v.y), h23 = pack_bf2(v.z, v.w);
          const uint32_t l01 = pack_bf2(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xffff0000u));
          const uint32_t l23 = pack_bf2(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xffff0000u));
          *(uint2*)(out_hi + grow * ld_split + spl_col(col)) = make_uint2(h01, h23);
          *(uint2*)(out_lo + grow * ld_split + spl_col(col)) = make_uint2(l01, l23);
        }
      }
  }
}
__global__ __launch_bounds__(256) void attention_x3_kernel(
    const uint16_t* __restrict__ qkv, int64_t ld, int q_col, int k_col, int v_col,
    const gims_attn_problem* __restrict__ problems, int n_groups, int n_heads, int n_qt, float* __restrict__ out,
    int64_t ld_out, uint16_t* __restrict__ out_hi, uint16_t* __restrict__ out_lo, int64_t ld_split, float c,
    unsigned long long* __restrict__ stat, gims_attn_guard guard, int n_blocks) {
  if (guard.stat) {          // guarded launch (include/gims_hip.h): the redo of a layer whose cheap tier did not suffice -- or nothing
    if (!attn_guard_fires(guard)) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) guard.stat[4 * guard.n_heads + 3] = 1ull;
  }
  for (int bid = (int)blockIdx.x; bid < n_blocks; bid += (int)gridDim.x) {      // (one tile per workgroup unless the launch is guarded: see attention_x3w_kernel)
    attention_x3_block(qkv, ld, q_col, k_col, v_col, problems, n_groups, n_heads, n_qt, out, ld_out, out_hi, out_lo, ld_split, c, stat, bid);
    if (bid + (int)gridDim.x < n_blocks) __syncthreads();
  }
}


// ---------------------------------------------------------------------------------------------- split-bf16 (x3), wide form
// attention_x3_kernel above gives a wave 32 queries and runs two workgroups per CU: per 64-key tile a wave reads 16 KB of K fragments and 16 KB of
// V^T fragments from LDS for 48 MFMAs -- LDS bandwidth, not the matrix pipes, bounds it (27 % MFMA occupancy, 1.2 ms per launch of 16 x 4096 keys).
// Here a wave owns 128 queries (four 32-query blocks), ONE wave per SIMD with the whole register file (<= 512 VGPRs), and walks its blocks in
// PAIRS: every K and V^T fragment serves two blocks (half the LDS bytes per MFMA), the 96 MFMAs of a pair are one stream in which the fragment
// reads of the next k-step sit between MFMAs that do not depend on them, and four waves stage a tile for 512 queries instead of 128 (a quarter of
// the staging stores and barriers per query).  Same arithmetic, same tile order and the same running-maximum rule per query block as
// attention_x3_kernel: the outputs are bit-identical to it.
constexpr int X3W_LDS_BYTES = X3_LDS_BYTES;
template <int X3W_QP>      // 32-query blocks per wave: 2 (two workgroups per CU, <= 256 registers; a QP = 4 instance with the whole register file was slower and left in round 6)
__device__ __forceinline__ void attention_x3w_block(
    const uint16_t* __restrict__ qkv, int64_t ld, int q_col, int k_col, int v_col,
    const gims_attn_problem* __restrict__ problems, int n_groups, int n_heads, int n_qt, float* __restrict__ out,
    int64_t ld_out, uint16_t* __restrict__ out_hi, uint16_t* __restrict__ out_lo, int64_t ld_split, float c,
    unsigned long long* __restrict__ stat, const int bid) {
  constexpr int X3W_QW = QW * X3W_QP, X3W_QB = X3W_QW * ATT_WAVES;
  extern __shared__ __attribute__((aligned(16))) uint16_t x3_lds[];
  auto Ks = [&](int p, int buf) __attribute__((always_inline)) { return x3_lds + (p * 2 + buf) * (KB * DH); };
  // V tiles ROW-MAJOR [key][d] with a 160-byte pitch (X3W_VR elements): filled by LDS-DMA like K, read transposed by ds_read_b64_tr_b16 (the
  // four key rows of a transposing read land on disjoint quarters of the 64 banks at this pitch, like the 192-byte pitch of attention8_bf16_kernel)
  auto Vr = [&](int p, int buf) __attribute__((always_inline)) { return x3_lds + 4 * (KB * DH) + (p * 2 + buf) * (KB * X3W_VR); };

  const int xcd = bid & 7, slot = bid >> 3;
  const int group = (slot / n_qt) * 8 + xcd;
  if (group >= n_groups) return;
  const gims_attn_problem pr = problems[group / n_heads];
  const int q0 = (slot % n_qt) * X3W_QB;
  if (q0 >= pr.n_q) return;
  const int head = group % n_heads;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int qs = spl_col(q_col + head * DH), ks = spl_col(k_col + head * DH), vs = spl_col(v_col + head * DH);
  auto doff = [&](int d8) __attribute__((always_inline)) { return ((d8 >> 2) << 6) + ((d8 & 3) << 3); };

  bf16x8 qh[X3W_QP][4], ql[X3W_QP][4];
#pragma unroll
  for (int qi = 0; qi < X3W_QP; ++qi) {
    int qr = q0 + wave * X3W_QW + qi * QW + li;
    qr = qr < pr.n_q ? qr : pr.n_q - 1;
    const uint16_t* qp = qkv + (int64_t)(pr.q_off + qr) * ld + qs;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qh[qi][s] = *(const bf16x8*)(qp + doff(2 * s + lh));
      ql[qi][s] = *(const bf16x8*)(qp + doff(2 * s + lh) + 32);
    }
  }
  f32x16 o[X3W_QP][2];
  float m_run[X3W_QP], l_run[X3W_QP], m_true[X3W_QP];
#pragma unroll
  for (int qi = 0; qi < X3W_QP; ++qi) {
    m_run[qi] = -1e30f; l_run[qi] = 0.f; m_true[qi] = -1e30f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qi][i][r] = 0.f;
  }
  constexpr float DEFER = 5.0f;
  const float defer_raw = DEFER / c;

  // staging registers of the NEXT tile, requested at the top of a tile and stored at its end (storing K behind the first pair's products and
  // requesting V only then -- one set of 8 registers live at a time -- measured slower: 937 -> 983 us)
  const int n_tiles = (pr.n_kv + KB - 1) / KB;
  // K goes global -> LDS by LDS-DMA (no staging registers: held across a tile they were what spilled -- 230 MB of scratch writes per launch in the
  // PMC pass).  A DMA instruction fills one contiguous KiB = 8 tile rows x 8 chunks; the chunk swizzle of k_off is applied on the SOURCE side:
  // the lane that lands at (row, position) fetches chunk position ^ ((row >> 1) & 7).  Two instructions per wave and plane.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto dma_k = [&](int kt, int buf) __attribute__((always_inline)) {
    const int kbase = kt * KB;
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      const int j = wave_u * 2 + j2, row = 8 * j + (lane >> 3), ch = (lane & 7) ^ ((row >> 1) & 7);
      int kr = kbase + row; kr = kr < pr.n_kv ? kr : pr.n_kv - 1;
      const uint16_t* src = qkv + (int64_t)(pr.kv_off + kr) * ld + ks + doff(ch);
#pragma unroll
      for (int p = 0; p < 2; ++p)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 32 * p),
                                         (__attribute__((address_space(3))) void*)(Ks(p, buf) + j * 512), 16, 0, 0);
    }
  };
  // a DMA instruction fills one contiguous KiB of the padded tile image: byte B = 1024 j + 16 lane is (row B / 160, chunk (B % 160) / 16); the
  // two pad chunks of a row re-fetch chunk 0.  Ten instructions per plane and tile: waves 0 and 1 issue three, waves 2 and 3 two.
  auto dma_v = [&](int kt, int buf) __attribute__((always_inline)) {
    const int kbase = kt * KB;
#pragma unroll
    for (int j3 = 0; j3 < 3; ++j3) {
      const int j = wave_u + 4 * j3;
      if (j < 10) {
        const int B = 1024 * j + 16 * lane, row = B / (2 * X3W_VR), pos = (B % (2 * X3W_VR)) >> 4, ch = pos < 8 ? pos : 0;
        int kr = kbase + row; kr = kr < pr.n_kv ? kr : pr.n_kv - 1;
        const uint16_t* src = qkv + (int64_t)(pr.kv_off + kr) * ld + vs + doff(ch);
#pragma unroll
        for (int p = 0; p < 2; ++p)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 32 * p),
                                           (__attribute__((address_space(3))) void*)(Vr(p, buf) + j * 512), 16, 0, 0);
      }
    }
  };
  dma_k(0, 0);
  dma_v(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < n_tiles; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < n_tiles) { dma_k(kt + 1, buf ^ 1); dma_v(kt + 1, buf ^ 1); }
    const int kbase = kt * KB;
#pragma unroll
    for (int pq = 0; pq < X3W_QP; pq += 2) {              // a pair of query blocks shares every K / V^T fragment
      f32x16 sacc[2][2];                                  // [block of the pair][key block]
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[u][b][r] = 0.f;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bf16x8 kh = *(const bf16x8*)(Ks(0, buf) + k_off(b * 32 + li, 2 * s + lh));
          const bf16x8 kl = *(const bf16x8*)(Ks(1, buf) + k_off(b * 32 + li, 2 * s + lh));
#pragma unroll
          for (int u = 0; u < 2; ++u) {                   // per accumulator: small terms first, as in attention_x3_kernel
            sacc[u][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[pq + u][s], sacc[u][b], 0, 0, 0);
            sacc[u][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[pq + u][s], sacc[u][b], 0, 0, 0);
            sacc[u][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[pq + u][s], sacc[u][b], 0, 0, 0);
          }
        }
      bf16x8 ph[2][4], pl[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qi = pq + u;
        if (kbase + KB > pr.n_kv) {
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = kbase + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
              if (key >= pr.n_kv) sacc[u][b][r] = -1e30f;
            }
        }
        float tmax = -1e30f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[u][b][r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        m_true[qi] = fmaxf(m_true[qi], tmax);
        if (__any(tmax > m_run[qi] + defer_raw)) {
          const float m_new = fmaxf(m_run[qi], tmax);
          const float alpha = __builtin_amdgcn_exp2f((m_run[qi] - m_new) * c);
          m_run[qi] = m_new;
          l_run[qi] *= alpha;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qi][i][r] *= alpha;
        }
        const float mc = m_run[qi] * c;
        float lsum = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float pv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            pv[r] = __builtin_amdgcn_exp2f(fmaf(sacc[u][b][r], c, -mc));
            lsum += pv[r];
          }
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            uint32_t hw[4], lw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float x = pv[8 * h2 + 2 * j], y = pv[8 * h2 + 2 * j + 1];
              hw[j] = pack_bf2(x, y);
              lw[j] = pack_bf2(x - __uint_as_float(hw[j] << 16), y - __uint_as_float(hw[j] & 0xffff0000u));
            }
            ph[u][2 * b + h2] = __builtin_bit_cast(bf16x8, make_uint4(hw[0], hw[1], hw[2], hw[3]));
            pl[u][2 * b + h2] = __builtin_bit_cast(bf16x8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
          }
        }
        l_run[qi] += lsum;
      }
      // (forming P one 16-key slice at a time, straight into that k-step's products -- 48 registers less on paper -- measured slower: 937 -> 1058 us)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          // V^T fragments by transposing reads of the row-major tiles (addressing of attention8_bf16_kernel's seg_pv: MFMA k-slot j of lane
          // (d = li, lh) is key 16 s + 8 (j >> 2) + 4 lh + (j & 3) -- the order P already has)
          const int voff = (4 * lh + ((lane & 15) >> 2)) * X3W_VR + 16 * ((lane >> 4) & 1) + 4 * (lane & 3) + 16 * s * X3W_VR + 32 * i;
          const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Vr(0, buf) + voff));
          const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Vr(0, buf) + voff + 8 * X3W_VR));
          const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Vr(1, buf) + voff));
          const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Vr(1, buf) + voff + 8 * X3W_VR));
          const bf16x8 vh = __builtin_bit_cast(bf16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
          const bf16x8 vl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            o[pq + u][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph[u][s], o[pq + u][i], 0, 0, 0);
            o[pq + u][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl[u][s], o[pq + u][i], 0, 0, 0);
            o[pq + u][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph[u][s], o[pq + u][i], 0, 0, 0);
          }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the DMA of the next K tile has landed before anyone crosses the barrier
    __syncthreads();
  }

#pragma unroll
  for (int qi = 0; qi < X3W_QP; ++qi) {
    const float l_tot = l_run[qi] + __shfl_xor(l_run[qi], 32, 64);
    const float inv = 1.f / l_tot;
    const int qr = q0 + wave * X3W_QW + qi * QW + li;
    if (stat) emit_peak_stat(stat, head, __builtin_amdgcn_exp2f((m_true[qi] - m_run[qi]) * c) * inv, lh == 0 && qr < pr.n_q);
    if (qr < pr.n_q) {
      const int64_t grow = pr.q_off + qr;
      const int col0 = head * DH + 4 * lh;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = make_float4(o[qi][i][4 * g] * inv, o[qi][i][4 * g + 1] * inv, o[qi][i][4 * g + 2] * inv, o[qi][i][4 * g + 3] * inv);
          const int col = col0 + 32 * i + 8 * g;
          if (out) *(float4*)(out + grow * ld_out + col) = v;
          if (out_hi) {
            const uint32_t h01 = pack_bf2(v.x, v.y), h23 = pack_bf2(v.z, v.w);
            const uint32_t l01 = pack_bf2(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xffff0000u));
            const uint32_t l23 = pack_bf2(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xffff0000u));
            *(uint2*)(out_hi + grow * ld_split + spl_col(col)) = make_uint2(h01, h23);
            *(uint2*)(out_lo + grow * ld_split + spl_col(col)) = make_uint2(l01, l23);
          }
        }
    }
  }
}
// One query tile per workgroup, or -- GUARDED launches that would need more than one dispatch round -- a strided walk over the tiles from ONE
// round of workgroups: a guarded launch that does not fire then costs one round of early exits instead of four (the device-side redo of
// attention_precision='auto' sits behind every bf16 / half layer of a match_pairs batch: 18 such launches per batch).  Same arithmetic per tile.
template <int X3W_QP, bool WALK>      // WALK: guarded launches only (the loop around the body costs the 256-register instance 48 more bytes of scratch per lane)
__global__ __launch_bounds__(256, X3W_QP == 2 ? 2 : 1) void attention_x3w_kernel(
    const uint16_t* __restrict__ qkv, int64_t ld, int q_col, int k_col, int v_col,
    const gims_attn_problem* __restrict__ problems, int n_groups, int n_heads, int n_qt, float* __restrict__ out,
    int64_t ld_out, uint16_t* __restrict__ out_hi, uint16_t* __restrict__ out_lo, int64_t ld_split, float c,
    unsigned long long* __restrict__ stat, gims_attn_guard guard, int n_blocks) {
  if (guard.stat) {          // guarded launch: see attention_x3_kernel
    if (!attn_guard_fires(guard)) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) guard.stat[4 * guard.n_heads + 3] = 1ull;
  }
  if constexpr (!WALK) {
    attention_x3w_block<X3W_QP>(qkv, ld, q_col, k_col, v_col, problems, n_groups, n_heads, n_qt, out, ld_out, out_hi, out_lo, ld_split, c, stat, (int)blockIdx.x);
  } else {
    for (int bid = (int)blockIdx.x; bid < n_blocks; bid += (int)gridDim.x) {
      attention_x3w_block<X3W_QP>(qkv, ld, q_col, k_col, v_col, problems, n_groups, n_heads, n_qt, out, ld_out, out_hi, out_lo, ld_split, c, stat, bid);
      if (bid + (int)gridDim.x < n_blocks) __syncthreads();      // the next tile's LDS-DMA overwrites the K / V buffers
    }
  }
}

// ---------------------------------------------------------------------------------------------- split-key variant (small launches)
// One image pair through the reference-shaped forward() (B = 1) gives a launch of 2 problems x 4 heads: with 32 queries per
// wave that is ONE wave per SIMD at 4096 keypoints (and a quarter of the chip at 1024) -- every wave then sits out its own
// LDS and global latencies.  Here a workgroup is eight waves over the same 128 queries: waves 0-3 walk the even half of the
// key tiles, waves 4-7 the other half (own K / V^T staging buffers), and the two partial results (unnormalised O, running
// maximum, row sum) are merged through LDS at the end -- the flash-decoding split, inside one workgroup: no workspace, no
// second launch.  Same math and layouts as attention_bf16_kernel<1>.
template <int NS, bool F16 = false>      // key parts per query block: 2 (eight waves) or 4 (sixteen waves); F16: operands and P in IEEE half
__global__ __launch_bounds__(256 * NS) void attention_split_kernel(
    const uint16_t* __restrict__ qkv, int64_t ld, int q_col, int k_col, int v_col,
    const gims_attn_problem* __restrict__ problems, int n_groups, int n_heads, int n_qt, float* __restrict__ out,
    int64_t ld_out, uint16_t* __restrict__ out_hi, uint16_t* __restrict__ out_lo, int64_t ld_split, float c,
    unsigned long long* __restrict__ stat) {
  extern __shared__ __attribute__((aligned(16))) uint16_t sp_lds[];
  __shared__ unsigned long long pacc[4];                                 // peakedness statistic of this workgroup (peak_acc_*)
  if (stat) peak_acc_init(pacc);
  // [half][buffer]: K tiles, then V^T tiles
  auto Ks = [&](int hf, int buf) __attribute__((always_inline)) { return sp_lds + (hf * 2 + buf) * (KB * DH); };
  auto Vr = [&](int hf, int buf) __attribute__((always_inline)) { return sp_lds + 2 * NS * (KB * DH) + (hf * 2 + buf) * (KB * VR_LD); };      // row-major V tiles

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int group = (slot / n_qt) * 8 + xcd;
  if (group >= n_groups) return;
  const gims_attn_problem pr = problems[group / n_heads];
  const int q0 = (slot % n_qt) * QB;
  if (q0 >= pr.n_q) return;
  const int head = group % n_heads;
  const int tt = threadIdx.x, lane = tt & 63;
  const int half = __builtin_amdgcn_readfirstlane(tt >> 8), wave = (tt >> 6) & 3;
  const int li = lane & 31, lh = lane >> 5;

  bf16x8 qf[4];
  {
    int qr = q0 + wave * QW + li;
    qr = qr < pr.n_q ? qr : pr.n_q - 1;
    const uint16_t* qp = qkv + (int64_t)(pr.q_off + qr) * ld + q_col + head * DH + 8 * lh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8*)(qp + 16 * s);
  }
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  constexpr float DEFER = 5.0f;
  const float defer_raw = DEFER / c;
  float m_run = -1e30f, l_run = 0.f, m_true = -1e30f;

  const int n_tiles = (pr.n_kv + KB - 1) / KB;
  const int n_mine = (n_tiles - half + NS - 1) / NS;      // tiles half, half + NS, ...
  const int n_iter = (n_tiles + NS - 1) / NS;             // every part passes the same number of barriers
  // K by LDS-DMA (see attention_bf16_kernel: its staging registers were spilled to scratch -- 48 / 128 B per lane here); V^T through registers
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto dma_k = [&](int it, int buf) __attribute__((always_inline)) {
    const int kbase = (NS * it + half) * KB;
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      const int j = wave_u * 2 + j2, row = 8 * j + (lane >> 3), ch = (lane & 7) ^ ((row >> 1) & 7);
      int kr = kbase + row; kr = kr < pr.n_kv ? kr : pr.n_kv - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qkv + (int64_t)(pr.kv_off + kr) * ld + k_col + head * DH + 8 * ch),
                                       (__attribute__((address_space(3))) void*)(Ks(half, buf) + j * 512), 16, 0, 0);
    }
  };
  auto dma_v = [&](int it, int buf) __attribute__((always_inline)) {      // row-major V tile of this part, 160-byte pitch (see attention_bf16_kernel)
    const int kbase = (NS * it + half) * KB;
#pragma unroll
    for (int j3 = 0; j3 < 3; ++j3) {
      const int j = wave_u + 4 * j3;
      if (j < 10) {
        const int B = 1024 * j + 16 * lane, row = B / (2 * VR_LD), pos = (B % (2 * VR_LD)) >> 4, ch = pos < 8 ? pos : 0;
        int kr = kbase + row; kr = kr < pr.n_kv ? kr : pr.n_kv - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qkv + (int64_t)(pr.kv_off + kr) * ld + v_col + head * DH + 8 * ch),
                                         (__attribute__((address_space(3))) void*)(Vr(half, buf) + j * 512), 16, 0, 0);
      }
    }
  };
  if (n_mine > 0) { dma_k(0, 0); dma_v(0, 0); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int it = 0; it < n_iter; ++it) {
    const int buf = it & 1;
    const bool live = it < n_mine;
    if (it + 1 < n_mine) { dma_k(it + 1, buf ^ 1); dma_v(it + 1, buf ^ 1); }
    if (live) {
      f32x16 sacc[2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[b][r] = 0.f;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bf16x8 kf = *(const bf16x8*)(Ks(half, buf) + k_off(b * 32 + li, 2 * s + lh));
          sacc[b] = mfma_16b<F16>(kf, qf[s], sacc[b]);
        }
      const int kbase = (NS * it + half) * KB;
      if (kbase + KB > pr.n_kv) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kbase + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (key >= pr.n_kv) sacc[b][r] = -1e30f;
          }
      }
      float tmax = -1e30f;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[b][r]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      m_true = fmaxf(m_true, tmax);
      if (__any(tmax > m_run + defer_raw)) {
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
      }
      const float mc = m_run * c;
      bf16x8 pf[4];
      float lsum = 0.f;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pv[r] = __builtin_amdgcn_exp2f(fmaf(sacc[b][r], c, -mc));
          lsum += pv[r];
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          uint4 pk;
          pk.x = pack_16b<F16>(pv[8 * h2 + 0], pv[8 * h2 + 1]);
          pk.y = pack_16b<F16>(pv[8 * h2 + 2], pv[8 * h2 + 3]);
          pk.z = pack_16b<F16>(pv[8 * h2 + 4], pv[8 * h2 + 5]);
          pk.w = pack_16b<F16>(pv[8 * h2 + 6], pv[8 * h2 + 7]);
          pf[2 * b + h2] = __builtin_bit_cast(bf16x8, pk);
        }
      }
      l_run += lsum;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int voff = (4 * lh + ((lane & 15) >> 2)) * VR_LD + 16 * ((lane >> 4) & 1) + 4 * (lane & 3) + 16 * s * VR_LD + 32 * i;
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Vr(half, buf) + voff));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Vr(half, buf) + voff + 8 * VR_LD));
          const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
          o[i] = mfma_16b<F16>(vf, pf[s], o[i]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the DMA of the next K tile has landed
    __syncthreads();
  }

  // ---- merge the parts: parts 1 .. NS-1 hand (O, m, l) to part 0 through LDS (the staging buffers are idle now)
  float* xo = (float*)sp_lds;                            // [NS - 1][4 waves][35][64 lanes]: 32 accumulator registers, m, l, row maximum
  if (half > 0) {
    float* dst = xo + (((half - 1) * 4 + wave) * 35) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(i * 16 + r) * 64] = o[i][r];
    dst[32 * 64] = m_run;
    dst[33 * 64] = l_run;
    dst[34 * 64] = m_true;
  }
  __syncthreads();
  if (half > 0) return;
#pragma unroll
  for (int pt = 1; pt < NS; ++pt) {                      // fixed merge order
    const float* src = xo + (((pt - 1) * 4 + wave) * 35) * 64 + lane;
    const float m1 = src[32 * 64], l1 = src[33 * 64];
    m_true = fmaxf(m_true, src[34 * 64]);
    const float m_new = fmaxf(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f((m_run - m_new) * c), a1 = __builtin_amdgcn_exp2f((m1 - m_new) * c);
    m_run = m_new;
    l_run = l_run * a0 + l1 * a1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] = o[i][r] * a0 + src[(i * 16 + r) * 64] * a1;
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  const int qr = q0 + wave * QW + li;
  if (stat) peak_acc_add(pacc, __builtin_amdgcn_exp2f((m_true - m_run) * c) * inv, lh == 0 && qr < pr.n_q);
  if (qr < pr.n_q) {
    const int64_t grow = pr.q_off + qr;
    const int col0 = head * DH + 4 * lh;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = make_float4(o[i][4 * g] * inv, o[i][4 * g + 1] * inv, o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv);
        const int col = col0 + 32 * i + 8 * g;
        if (out) *(float4*)(out + grow * ld_out + col) = v;
        if (out_hi) {
          const uint32_t h01 = pack_bf2(v.x, v.y), h23 = pack_bf2(v.z, v.w);
          const uint32_t l01 = pack_bf2(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xffff0000u));
          const uint32_t l23 = pack_bf2(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xffff0000u));
          *(uint2*)(out_hi + grow * ld_split + spl_col(col)) = make_uint2(h01, h23);
          *(uint2*)(out_lo + grow * ld_split + spl_col(col)) = make_uint2(l01, l23);
        }
      }
  }
  if (stat) peak_acc_flush(pacc, stat, head);
}
template <int NS> constexpr int SPLIT_LDS_BYTES = NS * 2 * (KB * DH + KB * VR_LD) * 2;     // >= the (NS - 1) x 35 KB of the merge exchange
static_assert(SPLIT_LDS_BYTES<2> >= 1 * 4 * 35 * 64 * 4 && SPLIT_LDS_BYTES<4> >= 3 * 4 * 35 * 64 * 4, "merge exchange must fit the staging buffers");

}  // namespace gims

namespace gims {
// |Q| (as stored: with the softmax scale when the caller folded it in), |K|, |V| abs-max of the rows a launch touches, for the range
// guard of the IEEE-half tier (finite range 65504): stat[n_heads][0..2] = the float bit patterns of the three maxima (positive floats
// order like their bit patterns: atomicMax on the integer view).  kind: 0 bf16 / 1 half buffers [rows][ld] of 16-bit values; 2 SPL32
// (the hi planes are read: |hi| is |x| to 2^-9).  16-byte pieces, grid-stride; launched on measured batches only.
__global__ __launch_bounds__(256) void attention_range_kernel(const uint16_t* __restrict__ qkv, int64_t ld, int q_col, int k_col, int v_col,
                                                              const gims_attn_problem* __restrict__ problems, int n_heads, int kind,
                                                              unsigned long long* __restrict__ stat) {
  const gims_attn_problem pr = problems[blockIdx.y];
  const int width = n_heads * DH;                      // logical channels per matrix
  const int cpr = width / 8;                           // 16-byte pieces per row and matrix
  float mx[3] = {0.f, 0.f, 0.f};
  auto fold = [&](int which, uint4 w) __attribute__((always_inline)) {
    const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a, b;
      if (kind == 1) {
        const f16x2 h = __builtin_bit_cast(f16x2, u[e]);
        a = fabsf((float)h[0]); b = fabsf((float)h[1]);
      } else {
        a = fabsf(__uint_as_float(u[e] << 16)); b = fabsf(__uint_as_float(u[e] & 0xffff0000u));
      }
      mx[which] = fmaxf(mx[which], fmaxf(a, b));        // (NaN operands are dropped by fmaxf; inf is kept)
    }
  };
#pragma unroll
  for (int which = 0; which < 3; ++which) {
    const int rows = which == 0 ? pr.n_q : pr.n_kv, off = which == 0 ? pr.q_off : pr.kv_off, col = which == 0 ? q_col : (which == 1 ? k_col : v_col);
    const int total = rows * cpr, stride = gridDim.x * blockDim.x;
    auto src = [&](int i) __attribute__((always_inline)) {
      const int r = i / cpr, ch = col + 8 * (i - r * cpr);
      return (const uint4*)(qkv + (int64_t)(off + r) * ld + (kind == 2 ? spl_col(ch) : ch));
    };
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < total; i += 4 * stride) {         // four 16-byte loads in flight per thread
      const uint4 w0 = *src(i), w1 = *src(i + stride), w2 = *src(i + 2 * stride), w3 = *src(i + 3 * stride);
      fold(which, w0); fold(which, w1); fold(which, w2); fold(which, w3);
    }
    for (; i < total; i += stride) fold(which, *src(i));
  }
  // one atomic per workgroup and quantity (thousands of waves on three addresses serialise: 150 us per launch with one atomic per wave)
  __shared__ float red[3][4];
#pragma unroll
  for (int which = 0; which < 3; ++which) {
    const float m = wave_max(mx[which]);
    if ((threadIdx.x & 63) == 0) red[which][threadIdx.x >> 6] = m;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const float m = fmaxf(fmaxf(red[threadIdx.x][0], red[threadIdx.x][1]), fmaxf(red[threadIdx.x][2], red[threadIdx.x][3]));
    // (a stale read can only be too small: the atomic then happens needlessly, never the other way round)
    unsigned long long* dst = stat + 4 * n_heads + threadIdx.x;
    if (m > 0.f && (unsigned long long)__float_as_uint(m) > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, (unsigned long long)__float_as_uint(m));
  }
}
static int attention_range_launch(const uint16_t* qkv, int64_t ld, int q_col, int k_col, int v_col, const gims_attn_problem* problems, int n_problems,
                                  int n_heads, int kind, unsigned long long* stat, hipStream_t stream) {
  hipLaunchKernelGGL(attention_range_kernel, dim3(32, n_problems), dim3(256), 0, stream, qkv, ld, q_col, k_col, v_col, problems, n_heads, kind, stat);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

// which kernel served a launch, counted per process (gims_attention_launch_counts: the parity tests assert that a batch ran on the
// kernel it is meant to pin -- the 8-wave kernel of the timed batches is only taken by launches that fill the chip)
static std::atomic<uint64_t> g_launch_counts[GIMS_ATTN_KERNEL_KINDS];
static inline void count_launch(int kind) { g_launch_counts[kind].fetch_add(1, std::memory_order_relaxed); }

// launch of the one-pass 16-bit kernels (F16 = false: bf16 operands, true: IEEE half), by launch shape
template <bool F16>
static int attention_16b_launch(const uint16_t* qkv, int64_t ld, int q_col, int k_col, int v_col, const gims_attn_problem* problems, int n_groups, int max_n_q,
                                int n_heads, float* out, int64_t ld_out, uint16_t* out_hi, uint16_t* out_lo, int64_t ld_split, bool prescaled, float c,
                                unsigned long long* stat, hipStream_t stream) {
  int force = 0;
  { const char* e = getenv("GIMS_ATTN_QP"); force = e ? atoi(e) : 0; }
  const int blocks2 = 8 * cdiv(n_groups, 8) * cdiv(max_n_q, 2 * QB);
  const bool two = force == 2 || (force != 1 && blocks2 >= 512);
  const int n_qt8 = cdiv(max_n_q, 512);
  const bool eight = force == 8 || (force == 0 && 8 * cdiv(n_groups, 8) * n_qt8 >= 256);   // 8-wave workgroups of 512 queries
  // a small launch (one pair through forward()): split the keys of every query block over two wave groups (GIMS_ATTN_QP=3: always)
  const bool split = force == 3 || (force == 0 && !eight && !two && 8 * cdiv(n_groups, 8) * cdiv(max_n_q, QB) <= 512 && max_n_q >= 512);
  if (split) {
    GIMS_LDS_ATTR((const void*)attention_split_kernel<2, F16>, SPLIT_LDS_BYTES<2>);
    GIMS_LDS_ATTR((const void*)attention_split_kernel<4, F16>, SPLIT_LDS_BYTES<4>);
    const int n_qt = cdiv(max_n_q, QB);
    const int wgs = 8 * cdiv(n_groups, 8) * n_qt;
    int ns_env = 0;                                  // read per call: the tests switch between the two variants
    { const char* e = getenv("GIMS_ATTN_SPLIT"); ns_env = e ? atoi(e) : 0; }
    // four key parts (sixteen waves) when the launch is at most one workgroup per CU and the keys are many
    const bool four = ns_env == 4 || (ns_env != 2 && wgs <= 256 && max_n_q >= 2048);
    count_launch(GIMS_ATTN_KERNEL_SPLIT);
    if (four)
      hipLaunchKernelGGL((attention_split_kernel<4, F16>), dim3(wgs), dim3(1024), SPLIT_LDS_BYTES<4>, stream, qkv, ld, q_col, k_col, v_col, problems,
                         n_groups, n_heads, n_qt, out, ld_out, out_hi, out_lo, ld_split, c, stat);
    else
      hipLaunchKernelGGL((attention_split_kernel<2, F16>), dim3(wgs), dim3(512), SPLIT_LDS_BYTES<2>, stream, qkv, ld, q_col, k_col, v_col, problems,
                         n_groups, n_heads, n_qt, out, ld_out, out_hi, out_lo, ld_split, c, stat);
  } else if (eight) {
    count_launch(F16 ? GIMS_ATTN_KERNEL_WAVE8_F16 : GIMS_ATTN_KERNEL_WAVE8);
    int exact_only = 0;                         // GIMS_ATTN_EXACT=1: running-maximum softmax only (no optimistic pass)
    { const char* e = getenv("GIMS_ATTN_EXACT"); exact_only = e ? atoi(e) : 0; }
    static int prof = -1;
    if (prof < 0) { const char* e = getenv("GIMS_ATTN_PROF"); prof = e ? atoi(e) : 0; }
    if (prof) {                                 // diagnostics only: synchronous, prints the phase anatomy of workgroup 0
      unsigned long long* dprof = (unsigned long long*)device_once("attention8_prof", 24 * sizeof(unsigned long long), nullptr);
      GIMS_CHECK_ARG(dprof, "gims_attention: no profile buffer");
      if (prescaled)
        hipLaunchKernelGGL((attention8_bf16_kernel<true, true, F16>), dim3(8 * cdiv(n_groups, 8) * n_qt8), dim3(512), 0, stream, qkv, ld,
                           q_col, k_col, v_col, problems, n_groups, n_heads, n_qt8, out, ld_out, out_hi, out_lo, ld_split, dprof, exact_only, c, nullptr, 0);
      else
        hipLaunchKernelGGL((attention8_bf16_kernel<true, false, F16>), dim3(8 * cdiv(n_groups, 8) * n_qt8), dim3(512), 0, stream, qkv, ld,
                           q_col, k_col, v_col, problems, n_groups, n_heads, n_qt8, out, ld_out, out_hi, out_lo, ld_split, dprof, exact_only, c, nullptr, 0);
      GIMS_HIP(hipStreamSynchronize(stream));
      unsigned long long h[24];
      GIMS_HIP(hipMemcpy(h, dprof, sizeof(h), hipMemcpyDeviceToHost));
      fprintf(stderr, "[attention8 wave 0 of workgroup 0] prologue %llu, tile loop %llu, epilogue %llu shader cycles; %.1f us on the 100-MHz counter -> %.2f GHz\n",
              h[20], h[21], h[22], h[23] / 100.0, (double)(h[20] + h[21] + h[22]) / (h[23] / 100.0) * 1e-3);
      for (int g = 0; g < 2; ++g)
        fprintf(stderr, "[attention8 wave %d] cycles over the whole tile loop: QK %llu, softmax0 %llu, softmax1 %llu, PV %llu, store_tile %llu, "
                "load_tile %llu, barrier wait %llu\n", 4 * g, h[g * 10 + 0], h[g * 10 + 2], h[g * 10 + 4], h[g * 10 + 8], h[g * 10 + 9], h[g * 10 + 6],
                h[g * 10 + 7]);
    } else {
      // the optimistic 8-wave kernel tracks no maximum: measured launches carry 8 * ceil(n_groups / 8) extra workgroups in front that measure
      // a sample of the rows (attention_peak_sample_wg)
      const int n_sample = stat ? 8 * cdiv(n_groups, 8) : 0;
      if (prescaled)
        hipLaunchKernelGGL((attention8_bf16_kernel<false, true, F16>), dim3(n_sample + 8 * cdiv(n_groups, 8) * n_qt8), dim3(512), 0, stream, qkv, ld,
                           q_col, k_col, v_col, problems, n_groups, n_heads, n_qt8, out, ld_out, out_hi, out_lo, ld_split, nullptr, exact_only, c, stat, n_sample);
      else
        hipLaunchKernelGGL((attention8_bf16_kernel<false, false, F16>), dim3(n_sample + 8 * cdiv(n_groups, 8) * n_qt8), dim3(512), 0, stream, qkv, ld,
                           q_col, k_col, v_col, problems, n_groups, n_heads, n_qt8, out, ld_out, out_hi, out_lo, ld_split, nullptr, exact_only, c, stat, n_sample);
    }
  } else if (two) {
    count_launch(GIMS_ATTN_KERNEL_WAVE4);
    const int n_qt = cdiv(max_n_q, 2 * QB);
    hipLaunchKernelGGL((attention_bf16_kernel<2, F16>), dim3(8 * cdiv(n_groups, 8) * n_qt), dim3(256), 0, stream, qkv, ld,
                       q_col, k_col, v_col, problems, n_groups, n_heads, n_qt, out, ld_out, out_hi, out_lo, ld_split, c, stat);
  } else {
    count_launch(GIMS_ATTN_KERNEL_WAVE4);
    const int n_qt = cdiv(max_n_q, QB);
    hipLaunchKernelGGL((attention_bf16_kernel<1, F16>), dim3(8 * cdiv(n_groups, 8) * n_qt), dim3(256), 0, stream, qkv, ld,
                       q_col, k_col, v_col, problems, n_groups, n_heads, n_qt, out, ld_out, out_hi, out_lo, ld_split, c, stat);
  }
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}
}  // namespace gims

extern "C" int gims_attention_launch_counts(uint64_t* counts, int32_t n, int32_t reset) {
  GIMS_CHECK_ARG(counts || n == 0, "gims_attention_launch_counts: null pointer");
  for (int i = 0; i < n && i < GIMS_ATTN_KERNEL_KINDS; ++i) counts[i] = gims::g_launch_counts[i].load(std::memory_order_relaxed);
  for (int i = GIMS_ATTN_KERNEL_KINDS; i < n; ++i) counts[i] = 0;
  if (reset) for (auto& c : gims::g_launch_counts) c.store(0, std::memory_order_relaxed);
  return GIMS_OK;
}

extern "C" int gims_attention(const uint16_t* qkv, int64_t ld, int32_t q_col, int32_t k_col, int32_t v_col,
                              const gims_attn_problem* problems, int32_t n_problems, int32_t max_n_q,
                              int32_t n_heads, float* out, int64_t ld_out, uint16_t* out_hi, uint16_t* out_lo,
                              int64_t ld_split, int32_t flags, void* stream) {
  return gims_attention_stat(qkv, ld, q_col, k_col, v_col, problems, n_problems, max_n_q, n_heads, out, ld_out, out_hi, out_lo, ld_split, flags,
                             nullptr, stream);
}

extern "C" int gims_attention_stat(const uint16_t* qkv, int64_t ld, int32_t q_col, int32_t k_col, int32_t v_col,
                                   const gims_attn_problem* problems, int32_t n_problems, int32_t max_n_q,
                                   int32_t n_heads, float* out, int64_t ld_out, uint16_t* out_hi, uint16_t* out_lo,
                                   int64_t ld_split, int32_t flags, uint64_t* stat_u64, void* stream) {
  gims_attn_args a = {};
  a.qkv = qkv; a.ld = ld; a.q_col = q_col; a.k_col = k_col; a.v_col = v_col; a.problems = problems; a.n_problems = n_problems; a.max_n_q = max_n_q;
  a.n_heads = n_heads; a.out = out; a.ld_out = ld_out; a.out_hi = out_hi; a.out_lo = out_lo; a.ld_split = ld_split; a.flags = flags; a.stat = stat_u64;
  return gims_attention_ex(&a, stream);
}

extern "C" int gims_attention_ex(const gims_attn_args* args, void* stream) {
  GIMS_CHECK_ARG(args, "gims_attention_ex: null arguments");
  const uint16_t* qkv = args->qkv;
  const int64_t ld = args->ld, ld_out = args->ld_out, ld_split = args->ld_split;
  const int32_t q_col = args->q_col, k_col = args->k_col, v_col = args->v_col, n_problems = args->n_problems, max_n_q = args->max_n_q, n_heads = args->n_heads,
                flags = args->flags;
  const gims_attn_problem* problems = args->problems;
  float* out = args->out;
  uint16_t* out_hi = args->out_hi;
  uint16_t* out_lo = args->out_lo;
  uint64_t* stat_u64 = args->stat;
  const gims_attn_guard guard = args->guard;
  GIMS_CHECK_ARG(!guard.stat || ((flags & GIMS_ATTN_X3) && !stat_u64 && (guard.kind == GIMS_GUARD_PEAKED || guard.kind == GIMS_GUARD_RANGE) &&
                                 guard.n_heads > 0 && guard.n_heads <= 15 && (((uintptr_t)guard.stat) & 7) == 0),
                 "gims_attention_ex: a guard goes with GIMS_ATTN_X3, without a statistic of its own, kind GIMS_GUARD_*, 8-byte aligned stat");
  using namespace gims;
  GIMS_CHECK_ARG((((uintptr_t)stat_u64) & 7) == 0, "gims_attention_stat: stat must be 8-byte aligned");
  unsigned long long* stat = (unsigned long long*)stat_u64;
  GIMS_CHECK_ARG(qkv && problems && (out || out_hi), "gims_attention: null pointer");
  GIMS_CHECK_ARG((out_hi == nullptr) == (out_lo == nullptr) && (ld_split % 8) == 0 && (((uintptr_t)out_hi | (uintptr_t)out_lo) & 15) == 0,
                 "gims_attention: out_hi/out_lo come together, 16-byte aligned, ld_split %% 8 == 0");
  GIMS_CHECK_ARG(n_problems > 0 && max_n_q > 0 && n_heads > 0, "gims_attention: empty launch");
  GIMS_CHECK_ARG((ld % 8) == 0 && (q_col % 8) == 0 && (k_col % 8) == 0 && (v_col % 8) == 0,
                 "gims_attention: qkv ld / column offsets must be multiples of 8 (16-byte loads)");
  GIMS_CHECK_ARG((ld_out % 4) == 0, "gims_attention: ld_out must be a multiple of 4");
  GIMS_CHECK_ARG(ld <= 16384, "gims_attention: qkv pitch %lld too large (32-bit tile offsets)", (long long)ld);
  const int n_groups = n_heads * n_problems;
  const bool prescaled = (flags & GIMS_ATTN_Q_PRESCALED) != 0;
  const float c = prescaled ? 1.f : 0.125f * 1.4426950408889634f;      // 1/sqrt(64) * log2(e)
  // 64 queries per wave (K/V fragments and barriers shared by two query blocks) when that still fills the chip
  // (environment read per call, not cached: the tests switch kernels with it)
  const bool f16 = (flags & GIMS_ATTN_F16) != 0;
  GIMS_CHECK_ARG(!(f16 && (flags & GIMS_ATTN_X3)), "gims_attention: GIMS_ATTN_F16 and GIMS_ATTN_X3 exclude each other");
  if (stat && (flags & (GIMS_ATTN_X3 | GIMS_ATTN_F16)) && !(flags & GIMS_ATTN_NO_RANGE)) {   // range of the operands as stored (measured launches of the half / calibration tiers: bf16 has f32's range)
    const int rc = attention_range_launch(qkv, ld, q_col, k_col, v_col, problems, n_problems, n_heads, (flags & GIMS_ATTN_X3) ? 2 : (f16 ? 1 : 0), stat,
                                          (hipStream_t)stream);
    if (rc != GIMS_OK) return rc;
  }
  if (flags & GIMS_ATTN_X3) {                  // split-bf16 operands from the SPL32 Q/K/V buffer, three MFMAs per product
    GIMS_CHECK_ARG((q_col % 32) == 0 && (k_col % 32) == 0 && (v_col % 32) == 0 && (ld % 64) == 0,
                   "gims_attention: GIMS_ATTN_X3 takes logical column offsets that are multiples of 32 and an SPL32 pitch (multiple of 64)");
    GIMS_LDS_ATTR((const void*)attention_x3_kernel, X3_LDS_BYTES);
    // wide form (64 queries per wave in a pair that shares every K / V fragment; 256-query workgroups, two per CU) when they fill the chip, else
    // the 32-query-per-wave kernel; GIMS_ATTN_X3W=0/2 forces.  Measured with K and V staged by LDS-DMA (32-query / wide; a 128-query-per-wave
    // form with the whole register file, QP = 4, was third everywhere -- 912 / 730 / 867 us at 16 x 4096 keys -- and left the library in round 6):
    // 32 x 2048 460 / 398, 40 x 1500 298 / 274, 64 x 1022 206 / 197, 8 x 700 26 / 41.
    int wide = -1;
    { const char* e = getenv("GIMS_ATTN_X3W"); if (e) wide = atoi(e); }
    if (wide < 0) wide = 8 * cdiv(n_groups, 8) * cdiv(max_n_q, 2 * QB) >= 512 ? 2 : 0;
    count_launch(guard.stat ? GIMS_ATTN_KERNEL_X3_GUARDED : GIMS_ATTN_KERNEL_X3);
    if (wide) {
      GIMS_LDS_ATTR((const void*)(attention_x3w_kernel<2, false>), X3W_LDS_BYTES);
      GIMS_LDS_ATTR((const void*)(attention_x3w_kernel<2, true>), X3W_LDS_BYTES);
      const int n_qtw = cdiv(max_n_q, 2 * QB);
      // guarded launches: one dispatch round of workgroups (two per CU), a strided walk over the tiles when the guard fires
      const int n_blocks = 8 * cdiv(n_groups, 8) * n_qtw, round = 2 * (device_cus() & ~7);
      if (guard.stat && n_blocks > round && guard_walk_enabled())
        hipLaunchKernelGGL((attention_x3w_kernel<2, true>), dim3(round), dim3(256), X3W_LDS_BYTES, (hipStream_t)stream, qkv, ld,
                           q_col, k_col, v_col, problems, n_groups, n_heads, n_qtw, out, ld_out, out_hi, out_lo, ld_split, c, stat, guard, n_blocks);
      else
        hipLaunchKernelGGL((attention_x3w_kernel<2, false>), dim3(n_blocks), dim3(256), X3W_LDS_BYTES, (hipStream_t)stream, qkv, ld,
                           q_col, k_col, v_col, problems, n_groups, n_heads, n_qtw, out, ld_out, out_hi, out_lo, ld_split, c, stat, guard, n_blocks);
      GIMS_LAUNCH_CHECK();
      return GIMS_OK;
    }
    const int n_qt = cdiv(max_n_q, QB);
    const int n_blocks = 8 * cdiv(n_groups, 8) * n_qt, round = 2 * (device_cus() & ~7);
    hipLaunchKernelGGL(attention_x3_kernel, dim3(guard.stat && n_blocks > round ? round : n_blocks), dim3(256), X3_LDS_BYTES, (hipStream_t)stream, qkv, ld,
                       q_col, k_col, v_col, problems, n_groups, n_heads, n_qt, out, ld_out, out_hi, out_lo, ld_split, c, stat, guard, n_blocks);
    GIMS_LAUNCH_CHECK();
    return GIMS_OK;
  }
  return f16 ? attention_16b_launch<true>(qkv, ld, q_col, k_col, v_col, problems, n_groups, max_n_q, n_heads, out, ld_out, out_hi, out_lo, ld_split, prescaled, c,
                                          stat, (hipStream_t)stream)
             : attention_16b_launch<false>(qkv, ld, q_col, k_col, v_col, problems, n_groups, max_n_q, n_heads, out, ld_out, out_hi, out_lo, ld_split, prescaled, c,
                                           stat, (hipStream_t)stream);
}


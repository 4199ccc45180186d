// DRAFT, NOT BUILT (kept for the record of the round-1 attention experiments).
// A 4-wave (one wave per SIMD, 512 registers) software-pipelined attention kernel: every wave interleaves its own MFMAs
// with its own softmax VALU, hand-ordered in groups of {1 MFMA, LDS reads 4 groups ahead, 7 VALU} fenced by sched_barrier.
// Correct (passed the attention tests incl. the optimistic-overflow fallback) but slower than attention8_bf16_kernel:
// 472 us vs 393 us per layer at 16 x 4096 (tools/attn_probe.py).  Cycle stamps per 64-key tile and wave: slot 1 1100,
// barrier 130, slot 2 1235, staging (4 ds_write_b128 + 4 global loads through registers) 800-1200.  The slots run at
// MFMA + VALU (no overlap to speak of once a group carries 7 VALU + 2 LDS instructions per MFMA: the issue budget behind
// one 32-cycle MFMA is ~5 instructions), and the register staging stalls on vmcnt one tile ahead.  Next steps if this is
// resumed: LDS-DMA staging 2-3 tiles ahead (the source-side swizzles are worked out in DESIGN.md), v_pk_add for the row
// sums, cross-slot fragment prefetch.  It was cut out of gims_amd/csrc/attention.hip; it needs that file's helpers.
// ---------------------------------------------------------------------------------------------- pipelined 8-wave variant
// What the phase probes (tools/probes/phase_overlap_probe.hip) showed about gfx950: the MFMAs of one wave and the VALU of
// its SIMD-mate do NOT overlap (16 MFMAs || one softmax block = 1056 cycles ~ 552 + 504), so attention8_bf16_kernel's
// "A multiplies while B exponentiates" phases run at the SUM of their parts.  A wave's OWN MFMAs and VALU do overlap when
// they alternate in its instruction stream (770 cycles per SIMD for the same work with two such waves per SIMD).  This
// kernel therefore gives every wave one software-pipelined stream with independent matrix and softmax work side by side:
//     slot 1 of key tile t:   MFMA  S1 = K_t Q1^T,  O1 += V_{t-1} P1(t-1)      VALU  P0(t) = softmax block 0 of tile t
//     barrier
//     slot 2 of key tile t:   MFMA  O0 += V_t P0(t),  S0 = K_{t+1} Q0^T          VALU  P1(t) = softmax block 1 of tile t
//                             + tile t+2 into LDS, request tile t+3
// (blocks 0/1 = the wave's two 32-query blocks).  Three K and three V tiles in LDS make one barrier per tile enough: the
// buffer written in slot 2 of tile t held tile t-1, last read in slot 1 of tile t.  The loop body is branch-free: the
// exponentials are referenced to the row maximum of the FIRST key tile (softmax is invariant to the reference; later
// scores above it give p > 1), the wrap-around terms are made harmless instead of skipped (P1(-1) = 0 against a zeroed V
// buffer; S0 of the tile after the last one is computed from stale LDS and never used), and only a ragged last tile takes
// the masking variant of the body.  A row sum beyond 1e30 (a score ~100 octaves above the reference) makes the workgroup
// repeat its tiles with attention_bf16_kernel<2>'s running-maximum code (same 256-query geometry).
template <int ILV>      // 0: product kernel; 1: the same with cycle stamps (GIMS_ATTN_PROF=1)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attention4p_bf16_kernel(
    const uint16_t* __restrict__ qkv, int64_t ld, int q_col, int k_col, int v_col,
    const gims_attn_problem* __restrict__ problems, int n_groups, int n_heads, int n_qt, float* __restrict__ out,
    int64_t ld_out, uint16_t* __restrict__ out_hi, uint16_t* __restrict__ out_lo, int64_t ld_split, unsigned long long* prof) {
  constexpr int VROW = 96;                       // V tile row-major, 192-byte pitch (see attention8_bf16_kernel)
  constexpr int KTILE = KB * DH, VTILE = KB * VROW;
  __shared__ __attribute__((aligned(16))) uint16_t Ks[3 * KTILE];
  __shared__ __attribute__((aligned(16))) uint16_t Vs[3 * VTILE];
  constexpr int QP = 2, QWV = QW * QP, QBK = QWV * 4;       // 64 queries per wave, 256 per workgroup

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int group = (slot / n_qt) * 8 + xcd;
  if (group >= n_groups) return;
  const gims_attn_problem pr = problems[group / n_heads];
  const int q0 = (slot % n_qt) * QBK;
  if (q0 >= pr.n_q) return;
  const int head = group % n_heads;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 31, lh = lane >> 5;

  bf16x8 qf[QP][4];
#pragma unroll
  for (int qi = 0; qi < QP; ++qi) {
    int qr = q0 + wave * QWV + qi * QW + li;
    qr = qr < pr.n_q ? qr : pr.n_q - 1;
    const uint16_t* qp = qkv + (int64_t)(pr.q_off + qr) * ld + q_col + head * DH + 8 * lh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[qi][s] = *(const bf16x8*)(qp + 16 * s);
  }
  const int n_tiles = (pr.n_kv + KB - 1) / KB;
  const float c = 0.125f * 1.4426950408889634f;

  // staging: a K and a V tile are 512 chunks of 16 bytes each; thread t carries chunks t and t + 256 of both
  uint4 rk[2], rv[2];
  auto load_tile = [&](int kt) __attribute__((always_inline)) {     // rows past the end re-read the last key: harmless
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int kr = kt * KB + 32 * h + (t >> 3); kr = kr < pr.n_kv ? kr : pr.n_kv - 1;
      const uint16_t* src = qkv + (int64_t)(pr.kv_off + kr) * ld + head * DH + 8 * (t & 7);
      rk[h] = *(const uint4*)(src + k_col);
      rv[h] = *(const uint4*)(src + v_col);
    }
  };
  const int kst = k_off(t >> 3, t & 7), vst = (t >> 3) * VROW + 8 * (t & 7);     // row + 32: same swizzle (k_off uses row bits 1-3)
  auto store_tile = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *(uint4*)(Ks + buf * KTILE + kst + 32 * h * DH) = rk[h];
      *(uint4*)(Vs + buf * VTILE + vst + 32 * h * VROW) = rv[h];
    }
  };
  // per-lane fragment bases; the tile buffer adds a wave-uniform offset, everything else is an immediate
  const int vfb = (4 * lh + ((lane & 15) >> 2)) * VROW + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  f32x16 o[QP][2], sacc[QP][2];
  uint32_t pfw[QP][16];                          // P as packed bf16 pairs: words 4s..4s+3 = the B fragment of PV k-step s
  float l_run[QP], mc[QP];
#pragma unroll
  for (int qi = 0; qi < QP; ++qi) {
    l_run[qi] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qi][i][r] = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) pfw[qi][w] = 0u;
  }

  auto mm_qk = [&](int qi, int buf) __attribute__((always_inline)) {       // sacc[qi] = K_buf Q_qi^T (prologue only)
    const uint16_t* kb = Ks + buf * KTILE;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[qi][b][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        sacc[qi][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(kb + k_off(b * 32 + li, 2 * s + lh)), qf[qi][s], sacc[qi][b], 0, 0, 0);
    }
  };
  auto mask_tail = [&](int qi, int kbase) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kbase + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (key >= pr.n_kv) sacc[qi][b][r] = -1e30f;
      }
  };
  auto raw_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // One slot = 16 groups of { 1 MFMA, the LDS reads of the MFMA PF groups ahead, 7 VALU of the softmax }, the order written
  // here being the order issued (sched_barrier after every group).  Matrix work on query block A: S_A = K[kbuf] Q_A^T (even
  // groups, chains b = 0, 1) and O_A += V[vbuf] P_A (odd groups, chains i = 0, 1); softmax on query block C: group m turns
  // the scores r = 2 (m & 7), +1 of key block m >> 3 into one packed word of P_C.  A's chains alternate, so dependent
  // MFMAs are two issues apart; nothing in a slot depends on anything else in it.
  constexpr int PF = 4;
  auto run_slot = [&](auto a_c, int kbuf, int vbuf) __attribute__((always_inline)) {
    constexpr int A = decltype(a_c)::value, C = 1 - A;
    const uint16_t* kb = Ks + kbuf * KTILE;
    const uint16_t* vb = Vs + vbuf * VTILE + vfb;
    bf16x8 fa[16];
    auto fetch = [&](int m) __attribute__((always_inline)) {
      const int j = m >> 1, blk = j >> 2, s = j & 3;
      if (m & 1) {
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vb + 16 * s * VROW + 32 * blk));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vb + (16 * s + 8) * VROW + 32 * blk));
        fa[m] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
      } else {
        fa[m] = *(const bf16x8*)(kb + k_off(blk * 32 + li, 2 * s + lh));
      }
    };
#pragma unroll
    for (int m = 0; m < PF; ++m) fetch(m);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[A][b][r] = 0.f;
    // the softmax chain of a score pair (fma -> exp -> add, pack) is spread over three consecutive groups, so no VALU of a
    // group waits for another one of the same group (one wave per SIMD: nobody else would fill the bubble)
    float lsum = 0.f;
    float x0[16], x1[16], e0[16], e1[16];
    x0[0] = fmaf(sacc[C][0][0], c, -mc[C]);
    x1[0] = fmaf(sacc[C][0][1], c, -mc[C]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (m + PF < 16) fetch(m + PF);
      const int j = m >> 1, blk = j >> 2, s = j & 3;
      if (m & 1)
        o[A][blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m], __builtin_bit_cast(bf16x8, make_uint4(pfw[A][4 * s], pfw[A][4 * s + 1], pfw[A][4 * s + 2], pfw[A][4 * s + 3])), o[A][blk], 0, 0, 0);
      else
        sacc[A][blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m], qf[A][s], sacc[A][blk], 0, 0, 0);
      e0[m] = __builtin_amdgcn_exp2f(x0[m]);
      e1[m] = __builtin_amdgcn_exp2f(x1[m]);
      if (m + 1 < 16) {
        const int bb = (m + 1) >> 3, r0 = 2 * ((m + 1) & 7);
        x0[m + 1] = fmaf(sacc[C][bb][r0], c, -mc[C]);
        x1[m + 1] = fmaf(sacc[C][bb][r0 + 1], c, -mc[C]);
      }
      if (m > 0) {
        lsum += e0[m - 1];
        lsum += e1[m - 1];
        pfw[C][m - 1] = pack_bf2(e0[m - 1], e1[m - 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    lsum += e0[15];
    lsum += e1[15];
    pfw[C][15] = pack_bf2(e0[15], e1[15]);
    l_run[C] += lsum;
  };

  // ---- prologue: tiles 0 and 1 into LDS, V buffer 2 zeroed (it plays tile "-1"), tile 2 requested; the reference
  // maxima of both query blocks from tile 0
  load_tile(0);
  for (int i = t; i < VTILE / 8; i += 256) *(uint4*)(Vs + 2 * VTILE + 8 * i) = make_uint4(0, 0, 0, 0);
  store_tile(0);
  load_tile(1);
  store_tile(1);
  load_tile(2);
  __syncthreads();
#pragma unroll
  for (int qi = 0; qi < QP; ++qi) {
    mm_qk(qi, 0);
    if (KB > pr.n_kv) mask_tail(qi, 0);
    float tmax = -1e30f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[qi][b][r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    mc[qi] = tmax * c;
  }
  // sacc[0] now holds S0 of tile 0 (masked if tile 0 is ragged), as slot 1 of tile 0 expects; slot 1 recomputes S1

  unsigned long long pacc[4] = {0, 0, 0, 0};
  const bool ragged = (pr.n_kv % KB) != 0;
  const int n_plain = ragged ? n_tiles - 1 : n_tiles;
  int b0 = 0, b1 = 1, b2 = 2;                    // (t, t+1, t+2) % 3
  auto body = [&](int kt, auto masked_c) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_c)::value;
    constexpr bool PROF = ILV == 1;              // the <1> instantiation carries cycle stamps (GIMS_ATTN_PROF=1)
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    if (PROF) t0 = __builtin_readcyclecounter();
    if (MASKED) mask_tail(0, kt * KB);           // S0 of this tile was computed unmasked in the previous slot 2
    run_slot(std::integral_constant<int, 1>{}, b0, b2);        // S1 = K_t Q1^T, O1 += V_{t-1} P1, P0 = softmax(S0)
    if (PROF) t1 = __builtin_readcyclecounter();
    raw_barrier();
    if (PROF) t2 = __builtin_readcyclecounter();
    if (MASKED) mask_tail(1, kt * KB);
    run_slot(std::integral_constant<int, 0>{}, b1, b0);        // S0 = K_{t+1} Q0^T, O0 += V_t P0, P1 = softmax(S1)
    if (PROF) t3 = __builtin_readcyclecounter();
    store_tile(b2);
    load_tile(kt + 3);
    if (PROF) { t4 = __builtin_readcyclecounter(); pacc[0] += t1 - t0; pacc[1] += t2 - t1; pacc[2] += t3 - t2; pacc[3] += t4 - t3; }
    const int nb = b0; b0 = b1; b1 = b2; b2 = nb;
  };
  for (int kt = 0; kt < n_plain; ++kt) body(kt, std::false_type{});
  if (ragged) {
    if (n_tiles == 1) { /* tile 0 was masked in the prologue already; masking again is idempotent */ }
    body(n_tiles - 1, std::true_type{});
  }
  if (ILV == 1 && prof && blockIdx.x == 0 && t == 0)
    for (int i = 0; i < 4; ++i) prof[i] = pacc[i];
  // drain: O1 += V_{n-1} P1(n-1)   (b2 is the buffer of the last tile after the rotation)
  raw_barrier();
  {
    const uint16_t* vb = Vs + b2 * VTILE + vfb;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vb + 16 * s * VROW + 32 * i));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vb + (16 * s + 8) * VROW + 32 * i));
        o[1][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7)),
                                                          __builtin_bit_cast(bf16x8, make_uint4(pfw[1][4 * s], pfw[1][4 * s + 1], pfw[1][4 * s + 2], pfw[1][4 * s + 3])), o[1][i], 0, 0, 0);
      }
  }

  bool bad = false;
#pragma unroll
  for (int qi = 0; qi < QP; ++qi) {
    l_run[qi] += __shfl_xor(l_run[qi], 32, 64);
    bad = bad || !(l_run[qi] < 1e30f);
  }
  if (__syncthreads_or(bad)) {                   // rare: redo with the running maximum
    attention_exact_body<2>(qkv, ld, q_col, k_col, v_col, problems, n_groups, n_heads, n_qt, out, ld_out, out_hi, out_lo, ld_split);
    return;
  }
#pragma unroll
  for (int qi = 0; qi < QP; ++qi) {
    const float inv = 1.f / l_run[qi];
    const int qr = q0 + wave * QWV + qi * QW + li;
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


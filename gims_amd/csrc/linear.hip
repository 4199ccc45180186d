// Linear layers / dense contractions on the gfx950 matrix cores.
//
//   C[m,n] = act(scale * sum_k A[m,k] W[n,k] + bias[n]) (+ R[m,n])        (see include/gims_hip.h)
//
// GIMS_PREC_F32   : v_mfma_f32_32x32x2_f32 -- exact f32 (bitwise a k-ordered fmaf chain), 157 TF peak.
// GIMS_PREC_BF16X3: v_mfma_f32_32x32x16_bf16 on split operands, hi*hi + hi*lo + lo*hi (~2^-17 relative),
//                   3 MFMAs per product at 16x the f32 rate.
//
// Tiling (both): 128x128 output tile per 256-thread workgroup, 2x2 waves, each wave a 64x64 sub-tile as
// 2x2 MFMA tiles of 32x32 (64 accumulator registers).  Operands are staged through LDS; the next K-tile
// is prefetched into registers while the current one is consumed (issue-early / write-late).
// Activations are point-major, so rows of A and of W are both K-contiguous: every global load is a full
// 128-byte line per row.
#include "common.h"

#include <stdlib.h>

namespace gims {

constexpr int BM = 128, BN = 128;

// ------------------------------------------------------------------------------------------ epilogue
__device__ __forceinline__ void epilogue_store(const gims_linear_args& p, int row, int col, float acc) {
  if (row >= p.m || col >= p.n) return;
  float v = acc * p.scale;
  if (p.bias) v += p.bias[col];
  if (p.act == GIMS_ACT_RELU) v = fmaxf(v, 0.f);
  if (p.residual) v += p.residual[(int64_t)row * p.ldc + col];
  if (p.out_f32) p.out_f32[(int64_t)row * p.ldc + col] = v;
  if (p.out_bf16) p.out_bf16[(int64_t)row * p.ldc_bf16 + col] = (p.flags & GIMS_LINEAR_OUT_F16) ? (uint16_t)(pack_h2_sat(v, 0.f) & 0xffffu) : f2bf(v);
  if (p.out_hi) {
    const uint16_t h = f2bf(v);
    p.out_hi[(int64_t)row * p.ld_split + spl_col(col)] = h;
    p.out_lo[(int64_t)row * p.ld_split + spl_col(col)] = f2bf(v - bf2f(h));
  }
}

// ------------------------------------------------------------------------------------------ f32 MFMA
// LDS image: As[k][m] / Ws[k][n] (k-major) so an MFMA operand read (lane -> row l&31, k = l>>5) is
// 32 consecutive floats per half-wave: conflict-free ds_read_b32.  Row pitch 129 makes the transposing
// ds_write_b32 of the staging pass conflict-free too (lane -> k-quad l&7, row l>>3).
constexpr int F32_BK = 32;
constexpr int F32_LD = BM + 1;

__device__ __forceinline__ void linear_f32_body(const gims_linear_args& p, float* As, float* Ws) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (m0 >= p.m || n0 >= p.n) return;   // batched launches are sized for the largest problem
  if ((p.flags & GIMS_LINEAR_UPPER) && n0 + BN <= m0) return;   // symmetric product: tile entirely below the diagonal
  const float* w = (const float*)p.w;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[4], rw[4];
  const int nk = p.k / F32_BK;

  auto load_tile = [&](int kt) {
    const int k = kt * F32_BK;
    const float* abase;
    int64_t lda;
    int kk;
    if (k < p.k0) { abase = p.a0; lda = p.lda0; kk = k; } else { abase = p.a1; lda = p.lda1; kk = k - p.k0; }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int f = t + 256 * it, row = f >> 3, kq = f & 7;
      int ar = m0 + row; ar = ar < p.m ? ar : p.m - 1;
      int wr = n0 + row; wr = wr < p.n ? wr : p.n - 1;
      ra[it] = *(const float4*)(abase + (int64_t)ar * lda + kk + 4 * kq);
      rw[it] = *(const float4*)(w + (int64_t)wr * p.ldw + k + 4 * kq);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int f = t + 256 * it, row = f >> 3, kq = f & 7;
      float* a = As + (4 * kq) * F32_LD + row;
      a[0] = ra[it].x; a[F32_LD] = ra[it].y; a[2 * F32_LD] = ra[it].z; a[3 * F32_LD] = ra[it].w;
      float* b = Ws + (4 * kq) * F32_LD + row;
      b[0] = rw[it].x; b[F32_LD] = rw[it].y; b[2 * F32_LD] = rw[it].z; b[3 * F32_LD] = rw[it].w;
    }
  };

  load_tile(0);
  store_tile();
  __syncthreads();
  const int li = lane & 31, lh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int s = 0; s < F32_BK / 2; ++s) {
      const float* ap = As + (2 * s + lh) * F32_LD + wm * 64 + li;
      const float* bp = Ws + (2 * s + lh) * F32_LD + wn * 64 + li;
      const float a0 = ap[0], a1 = ap[32], b0 = bp[0], b1 = bp[32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
    if (kt + 1 < nk) {
      store_tile();
      __syncthreads();
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int col = n0 + wn * 64 + j * 32 + li;
        epilogue_store(p, row, col, acc[i][j][r]);
      }
}

__global__ __launch_bounds__(256) void linear_f32_kernel(gims_linear_args p) {
  __shared__ float As[F32_BK * F32_LD];
  __shared__ float Ws[F32_BK * F32_LD];
  linear_f32_body(p, As, Ws);
}
// one launch, many independent problems (blockIdx.z): descriptors live in device memory
__global__ __launch_bounds__(256) void linear_f32_batch_kernel(const gims_linear_args* __restrict__ args) {
  __shared__ float As[F32_BK * F32_LD];
  __shared__ float Ws[F32_BK * F32_LD];
  const gims_linear_args p = args[blockIdx.z];
  linear_f32_body(p, As, Ws);
}

// ------------------------------------------------------------------------------------------ split-bf16 MFMA
// LDS image per plane: [128 rows][64 k] bf16 = 128-byte rows, 16-byte chunks XOR-swizzled with
// (row>>1)&7 so the ds_read_b128 of an MFMA operand (16 lanes of a group -> 16 different rows, same
// logical chunk) spreads over all 16 slots of the 256-byte bank line.
constexpr int X3_BK = 64;
constexpr int X3_PLANE = BM * X3_BK;  // bf16 elements per plane

__device__ __forceinline__ int x3_off(int row, int chunk) {  // element offset of a 16-byte chunk
  return row * X3_BK + ((chunk ^ ((row >> 1) & 7)) << 3);
}

__device__ __forceinline__ void linear_bf16x3_body(const gims_linear_args& p, uint16_t* smem) {
  uint16_t* Ah = smem;
  uint16_t* Al = smem + X3_PLANE;
  uint16_t* Wh = smem + 2 * X3_PLANE;
  uint16_t* Wl = smem + 3 * X3_PLANE;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (m0 >= p.m || n0 >= p.n) return;
  const uint16_t* wh = (const uint16_t*)p.w;
  const uint16_t* wl = (const uint16_t*)p.w_lo;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[8];
  uint4 rwh[4], rwl[4];
  const int nk = p.k / X3_BK;

  auto load_tile = [&](int kt) {
    const int k = kt * X3_BK;
    const float* abase;
    int64_t lda;
    int kk;
    if (k < p.k0) { abase = p.a0; lda = p.lda0; kk = k; } else { abase = p.a1; lda = p.lda1; kk = k - p.k0; }
#pragma unroll
    for (int it = 0; it < 8; ++it) {  // A: 128 rows x 16 float4
      const int f = t + 256 * it, row = f >> 4, kq = f & 15;
      int ar = m0 + row; ar = ar < p.m ? ar : p.m - 1;
      ra[it] = *(const float4*)(abase + (int64_t)ar * lda + kk + 4 * kq);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {  // W planes: 128 rows x 8 chunks of 8 bf16
      const int f = t + 256 * it, row = f >> 3, ch = f & 7;
      int wr = n0 + row; wr = wr < p.n ? wr : p.n - 1;
      rwh[it] = *(const uint4*)(wh + (int64_t)wr * p.ldw + k + 8 * ch);
      rwl[it] = *(const uint4*)(wl + (int64_t)wr * p.ldw + k + 8 * ch);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int f = t + 256 * it, row = f >> 4, kq = f & 15;
      const float4 x = ra[it];
      const uint32_t h01 = pack_bf2(x.x, x.y), h23 = pack_bf2(x.z, x.w);
      const uint32_t l01 = pack_bf2(x.x - __uint_as_float(h01 << 16), x.y - __uint_as_float(h01 & 0xffff0000u));
      const uint32_t l23 = pack_bf2(x.z - __uint_as_float(h23 << 16), x.w - __uint_as_float(h23 & 0xffff0000u));
      const int off = x3_off(row, kq >> 1) + 4 * (kq & 1);
      *(uint2*)(Ah + off) = make_uint2(h01, h23);
      *(uint2*)(Al + off) = make_uint2(l01, l23);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int f = t + 256 * it, row = f >> 3, ch = f & 7;
      *(uint4*)(Wh + x3_off(row, ch)) = rwh[it];
      *(uint4*)(Wl + x3_off(row, ch)) = rwl[it];
    }
  };

  load_tile(0);
  store_tile();
  __syncthreads();
  const int li = lane & 31, lh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int s = 0; s < X3_BK / 16; ++s) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ar = wm * 64 + i * 32 + li, br = wn * 64 + i * 32 + li;
        ah[i] = *(const bf16x8*)(Ah + x3_off(ar, 2 * s + lh));
        al[i] = *(const bf16x8*)(Al + x3_off(ar, 2 * s + lh));
        bh[i] = *(const bf16x8*)(Wh + x3_off(br, 2 * s + lh));
        bl[i] = *(const bf16x8*)(Wl + x3_off(br, 2 * s + lh));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // small terms first, then the dominant hi*hi product
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (kt + 1 < nk) {
      store_tile();
      __syncthreads();
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int col = n0 + wn * 64 + j * 32 + li;
        epilogue_store(p, row, col, acc[i][j][r]);
      }
}

__global__ __launch_bounds__(256, 2) void linear_bf16x3_kernel(gims_linear_args p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  linear_bf16x3_body(p, smem);
}
__global__ __launch_bounds__(256, 2) void linear_bf16x3_batch_kernel(const gims_linear_args* __restrict__ args) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  const gims_linear_args p = args[blockIdx.z];
  linear_bf16x3_body(p, smem);
}
// ------------------------------------------------------------------------------------------ split-bf16 MFMA, pre-split A
// The hot linears of the attentional GNN.  Activations and weights arrive ALREADY split, in the SPL32 layout
// (common.h): per row and 32-channel block one 128-byte line [32 hi | 32 lo], written by the producing kernel's
// epilogue.  Operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write):
//   * a K-step is one 32-channel block: every DMA piece (1 KiB) moves 8 FULL cache lines (hi and lo of 8 rows), so
//     the L2 sees half the requests of a plane-per-buffer layout with 64-byte rows (measured: that variant was
//     pinned at ~5.7 TB/s of L2->LDS traffic whatever the tile size);
//   * S-stage LDS ring, ONE raw s_barrier per K-step, counted s_waitcnt vmcnt so the DMA queue need not drain;
//   * the LDS image is lane-linear per piece, so the bank-conflict swizzle (16-byte chunk ^= (row>>1)&7) is applied
//     to the per-lane SOURCE address and again on the ds_read_b128 (same involution on both sides).
// The MFMA is issued with swapped operands (D^T = W A^T): a lane then owns 4 CONSECUTIVE output channels of
// one row per accumulator group, so every epilogue store is 8-16 bytes wide (f32x4 / bf16x4) and bias /
// residual are vector loads.
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// Geometry: TM x TN output tile per workgroup, WM x WN waves, each wave (TM/WM) x (TN/WN) as 32x32 MFMA tiles,
// 32-channel stages in an S-stage LDS ring.  Two instantiations are used:
//   128x128, 2x2 waves (64x64 per wave),  64 KiB LDS, 2 workgroups/CU -- small problems (fills the chip sooner);
//   256x256, 4x2 waves (64x128 per wave), 128 KiB LDS, 1 workgroup/CU -- half the operand bytes per MFMA, used when
//   the launch still yields at least ~one workgroup per CU.
template <int TM, int TN, int WM, int WN, int S, bool HALF = false>     // HALF: the hi halves of the rows only (64-byte rows)
struct X3P {
  static constexpr int BK = 32;
  static constexpr int WAVES = WM * WN;
  static constexpr int MI = TM / WM / 32, NI = TN / WN / 32;     // 32x32 tiles per wave
  static constexpr int ROWE = HALF ? 32 : 64;                    // bf16 elements per row of a stage (64- / 128-byte rows)
  static constexpr int RPP = 512 / ROWE;                          // rows per 1 KiB DMA piece (16 / 8)
  static constexpr int A_TILE = TM * ROWE, W_TILE = TN * ROWE;   // bf16 elements per stage
  static constexpr int STAGE = A_TILE + W_TILE;
  static constexpr int PA = TM / RPP, PW = TN / RPP;             // 1 KiB pieces per operand per stage
  static constexpr int PIECES = (PA + PW) / WAVES;               // pieces per wave per stage
  static constexpr int EPW = TN / WN > 128 ? 128 : TN / WN;       // columns of a wave tile that go through the epilogue transpose at a time
  static constexpr int EP_PITCH = EPW + 4;                        // epilogue transpose slice: 32 rows x EP_PITCH floats per wave
  static constexpr int RING_BYTES = S * STAGE * 2, EP_BYTES = WAVES * 32 * EP_PITCH * 4;
  static constexpr int LDS_BYTES = RING_BYTES > EP_BYTES ? RING_BYTES : EP_BYTES;
  static_assert((PA + PW) % WAVES == 0, "pieces divide evenly over the waves");
  // 16-byte chunk swizzles that make the fragment reads (32 rows x one chunk) conflict-free: 128-byte rows rotate by
  // row/2 over 8 chunks, 64-byte rows by row/4 over 4 chunks (four consecutive rows already cover the 256-byte bank row)
  __device__ static __forceinline__ int swz(int row) { return HALF ? (row >> 2) & 3 : (row >> 1) & 7; }
  __device__ static __forceinline__ int off(int row, int chunk) { return row * ROWE + ((chunk ^ swz(row)) << 3); }
};

// HI_ONLY: 1 = GIMS_LINEAR_HI_ONLY (one MFMA pass, hi planes), 4 = GIMS_LINEAR_CONV3 (all three passes; the A rows are gathered from the 3x3 neighbourhood of an NHWC activation)
// The body of one output tile (`bid` = the tile's slot in the XCD-aware order).  linear_x3p_kernel runs it once per workgroup; the
// persistent form below walks a strided list of tiles.
template <int TM, int TN, int WM, int WN, int S, int HI_ONLY>
__device__ __forceinline__ void linear_x3p_tile(const gims_linear_args& p, const int bid) {
  using T = X3P<TM, TN, WM, WN, S, HI_ONLY == 1>;
  constexpr int BK = T::BK;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order: workgroup b is dispatched to XCD b % 8 (observed; used for speed only), and each XCD has
  // its own L2.  All column tiles of one row panel of A are therefore given consecutive slots on ONE XCD, so the
  // panel is fetched from HBM once instead of once per column tile.
  const int nt_n = (p.n + TN - 1) / TN;
  const int xcd = bid & 7, slot = bid >> 3;
  const int m0 = ((slot / nt_n) * 8 + xcd) * TM, n0 = (slot % nt_n) * TN;
  if (m0 >= p.m) return;
  // probe (tools/gemm_probe.py): flag 0x1000 starts the workgroups of the odd slots conv_reserved x 3.4 us late, so that the
  // HBM-bound epilogues of one half of the CUs fall into the L2-bound K loops of the other half
  if ((p.flags & 0x1000) && (slot & 1))
    for (int d = 0; d < p.conv_reserved; ++d) __builtin_amdgcn_s_sleep(127);
  const int li = lane & 31, lh = lane >> 5;
  // diagnostic bits (tools/gemm_probe.py only): 0x100 = no main loop, 0x200 = no epilogue memory traffic, 0x400 / 0x800 / 0x2000 below
  const int nk = (p.flags & 0x100) ? 0 : p.k / BK;

  // this wave's LDS-DMA duty: PIECES consecutive 8-row pieces of the stage image [A rows | W rows]
  const int p0 = wave * T::PIECES;
  constexpr int CPR = T::ROWE / 8;                         // 16-byte chunks per stage row (8, or 4 for the hi-only rows)
  const int drow = lane / CPR, dpos = lane % CPR;
  // conv mode: this lane's A rows are output pixels; their input coordinates are fixed over the K loop
  constexpr bool CONV = HI_ONLY == 4;
  int cy[T::PIECES], cx[T::PIECES];
  int64_t cb[T::PIECES];
  const int cC = CONV ? p.k / 9 : 1;
  if (CONV) {
    const int wo = (p.conv_w - 1) / p.conv_stride + 1, ho = (p.conv_h - 1) / p.conv_stride + 1;
#pragma unroll
    for (int i = 0; i < T::PIECES; ++i) {
      int gr = m0 + T::RPP * (p0 + i) + drow;
      gr = gr < p.m - 1 ? gr : p.m - 1;
      const int t2 = gr / wo, xo = gr - t2 * wo, pi = t2 / ho, yo = t2 - pi * ho;
      cy[i] = yo * p.conv_stride; cx[i] = xo * p.conv_stride; cb[i] = (int64_t)pi * p.conv_h * p.conv_w;
    }
  }
  // Loop-invariant part of every DMA source address: the byte offset of this lane's 16-byte chunk inside the tile's row
  // panel (rows clamped to the last valid one), one per K segment.  The K-dependent part is a wave-uniform base pointer
  // (scalar ALU), so issuing a stage costs no vector ALU work -- VALU issue of one wave stalls the MFMAs of its SIMD mate,
  // and the 64-bit row x pitch multiplies this replaces were 17 % of the K loop.
  uint32_t voff0[T::PIECES], voff1[T::PIECES];
#pragma unroll
  for (int i = 0; i < T::PIECES; ++i) {
    const int pp = p0 + i;
    const bool is_a = pp < T::PA;
    const int row = T::RPP * (is_a ? pp : pp - T::PA) + drow;
    const int rmax = (is_a ? p.m - m0 : p.n - n0) - 1;
    const uint32_t rl = (uint32_t)(row < rmax ? row : rmax), ch = 16u * (uint32_t)(dpos ^ T::swz(row));
    voff0[i] = rl * (uint32_t)(is_a ? p.lda0 : p.ldw) * 2u + ch;
    voff1[i] = is_a ? rl * (uint32_t)p.lda1 * 2u + ch : voff0[i];
  }
  auto issue = [&](int kt) {
    const int k = kt * BK;
    const bool second = k >= p.k0;
    const int akk = second ? k - p.k0 : k;
    // SPL32: block k/32 starts at element 2*k
    const char* ab = (const char*)(second ? p.a1 : p.a0) + ((int64_t)m0 * (second ? p.lda1 : p.lda0) + 2 * akk) * 2;
    const char* wb = (const char*)p.w + ((int64_t)n0 * p.ldw + 2 * k) * 2;
    uint16_t* dst = smem + (kt % S) * T::STAGE + p0 * 512;
#pragma unroll
    for (int i = 0; i < T::PIECES; ++i) {
      const int pp = p0 + i;                               // wave-uniform
      const bool is_a = pp < T::PA;
      const char* g = (is_a ? ab : wb) + (second ? voff1[i] : voff0[i]);
      if (CONV && is_a) {                                  // tap (ky, kx) and channel block of this K step; zeros outside the image
        const int row = T::RPP * pp + drow;
        const int tap = k / cC, c0 = k - tap * cC;
        const int y = cy[i] + tap / 3 - 1, x = cx[i] + tap % 3 - 1;
        const bool inb = y >= 0 && y < p.conv_h && x >= 0 && x < p.conv_w;
        g = (const char*)(inb ? (const uint16_t*)p.a0 + (cb[i] + (int64_t)y * p.conv_w + x) * p.lda0 + 2 * c0 + 8 * (dpos ^ T::swz(row))
                              : (const uint16_t*)p.a1 + 8 * dpos);
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(dst + i * 512), 16, 0, 0);
    }
  };

  f32x16 acc[T::NI][T::MI];   // [n-block][m-block], D^T layout: column = row m (lane&31), rows = output channels
#pragma unroll
  for (int i = 0; i < T::NI; ++i)
#pragma unroll
    for (int j = 0; j < T::MI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  constexpr int D = S - 1;   // stages in flight
#pragma unroll
  for (int s0 = 0; s0 < D; ++s0)
    if (s0 < nk) issue(s0);
  for (int kt = 0; kt < nk; ++kt) {
    // own pieces of stage kt landed: at most `newer` stages (PIECES each) may still be in flight
    const int newer = (nk - 1 - kt) < (D - 1) ? (nk - 1 - kt) : (D - 1);
    if (newer == 0) wait_vm<0>();
    else if (newer >= 2) wait_vm<2 * T::PIECES>();
    else wait_vm<T::PIECES>();
    __builtin_amdgcn_s_barrier();
    if (kt + D < nk && !(p.flags & 0x400)) issue(kt + D);     // 0x400: no DMA after the prologue (probe)
    if (p.flags & 0x800) continue;                             // 0x800: no LDS reads / MFMAs (probe)
    const uint16_t* st = smem + (kt % S) * T::STAGE;
    // Fragment reads are grouped and pinned ahead of the MFMAs that consume them (in consumption order: LDS returns in
    // order, so the waits count down): reads of K step 0 -> its lo*hi and hi*lo products -> reads of step 1 -> hi*hi of
    // step 0 -> step 1.  Left alone the compiler emits read-a-few / wait-for-ALL / multiply-a-few, eight exposed LDS round
    // trips per stage; more than 15 reads in flight cannot be counted by lgkmcnt either, hence two groups.
    constexpr bool LO = HI_ONLY != 1;
    const bool lo_pass = HI_ONLY == 0 || HI_ONLY == 4;
    bf16x8 ah[2][T::MI], al[2][T::MI], wh[2][T::NI], wl[2][T::NI];
    auto rd = [&](int s) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < T::MI; ++i) ah[s][i] = *(const bf16x8*)(st + T::off(wm * (TM / WM) + i * 32 + li, 2 * s + lh));       // chunks 0-3: hi of channels 0-31
      if (LO) {
#pragma unroll
        for (int i = 0; i < T::NI; ++i) wl[s][i] = *(const bf16x8*)(st + T::A_TILE + T::off(wn * (TN / WN) + i * 32 + li, 4 + 2 * s + lh));
#pragma unroll
        for (int i = 0; i < T::MI; ++i) al[s][i] = *(const bf16x8*)(st + T::off(wm * (TM / WM) + i * 32 + li, 4 + 2 * s + lh));   // chunks 4-7: lo
      }
#pragma unroll
      for (int i = 0; i < T::NI; ++i) wh[s][i] = *(const bf16x8*)(st + T::A_TILE + T::off(wn * (TN / WN) + i * 32 + li, 2 * s + lh));
    };
    // term-major order: consecutive MFMAs hit DIFFERENT accumulators (no back-to-back dependent issue); per
    // accumulator the order stays lo*hi, hi*lo, hi*hi (small terms first)
    auto lo_terms = [&](int s) __attribute__((always_inline)) {
      if (LO && lo_pass) {
#pragma unroll
        for (int ni = 0; ni < T::NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < T::MI; ++mi) acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[s][ni], ah[s][mi], acc[ni][mi], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < T::NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < T::MI; ++mi) acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s][ni], al[s][mi], acc[ni][mi], 0, 0, 0);
      }
    };
    auto hi_term = [&](int s) __attribute__((always_inline)) {
#pragma unroll
      for (int ni = 0; ni < T::NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < T::MI; ++mi) acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s][ni], ah[s][mi], acc[ni][mi], 0, 0, 0);
    };
    static_assert(BK == 32, "two 16-deep K steps per stage");
    const bool no_rd = p.flags & 0x2000;                       // 0x2000: MFMAs on whatever the fragment registers hold, no LDS reads (probe)
    if (!no_rd) rd(0);
    if (!LO && !no_rd) rd(1);
    __builtin_amdgcn_sched_barrier(0);
    lo_terms(0);
    __builtin_amdgcn_sched_barrier(0);
    if (LO && !no_rd) rd(1);
    __builtin_amdgcn_sched_barrier(0);
    hi_term(0);
    lo_terms(1);
    hi_term(1);
  }

  // ---- epilogue.  In the MFMA (D^T) layout a lane owns 4 consecutive channels of ONE row, so a wave-wide store would
  // scatter 16-byte pieces over 32 rows; the memory system then sees 4-8x the transactions of a row-contiguous store and
  // the epilogue becomes transaction-bound (measured: ~45 us per launch at 65k rows whatever the output bytes).  Each
  // wave therefore transposes its accumulators through a private slice of the (now idle) LDS ring, 32 rows at a time,
  // and reads them back row-contiguous: a wave-wide access then covers whole 128-byte lines of bias / residual / outputs.
  if ((p.flags & 0x200) && acc[0][0][0] != 12345.678f) return;
  __builtin_amdgcn_s_barrier();                       // every wave is done reading the ring
  constexpr int WCOLS = TN / WN;                      // columns of the wave tile
  constexpr int EPW = T::EPW, NG = WCOLS / EPW, NIG = EPW / 32;   // ... transposed EPW columns (NIG 32-column blocks) at a time
  constexpr int PITCH = T::EP_PITCH;                  // floats; +4 keeps the 16-byte LDS writes of 16 lanes on distinct banks
  constexpr int LPR = EPW / 8;                        // lanes per row in the read-back (8 consecutive channels per lane)
  constexpr int RPI = 64 / LPR;                       // rows per wave-wide access
  constexpr int ITERS = 32 / RPI;
  float* ep = (float*)smem + wave * (32 * PITCH);
  const int c8 = (lane % LPR) * 8, rsub = lane / LPR;
  float rmax[3] = {0.f, 0.f, 0.f};                    // GIMS_LINEAR_OUT_F16 + range_stat: max |stored value| per 256-column block (Q | K | V)
#pragma unroll
  for (int mi = 0; mi < T::MI; ++mi) {
#pragma unroll
   for (int cg = 0; cg < NG; ++cg) {
    const int col = n0 + wn * WCOLS + cg * EPW + c8;
    const bool cok = col < p.n, full = col + 4 < p.n;   // n % 4 == 0: a lane's 8 channels are all, half or not in range
    float4 b4[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) b4[h] = (p.bias && col + 4 * h < p.n) ? *(const float4*)(p.bias + col + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int scol = spl_col(col);
#pragma unroll
    for (int ni = 0; ni < NIG; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(float4*)(ep + li * PITCH + ni * 32 + 8 * g + 4 * lh) =
            make_float4(acc[cg * NIG + ni][mi][4 * g], acc[cg * NIG + ni][mi][4 * g + 1], acc[cg * NIG + ni][mi][4 * g + 2], acc[cg * NIG + ni][mi][4 * g + 3]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private slice: no workgroup barrier needed
    const int rbase = m0 + wm * (TM / WM) + mi * 32 + rsub;
    constexpr int CH = ITERS < 4 ? ITERS : 4;            // residual loads of a chunk are all in flight before the first use
#pragma unroll
    for (int it0 = 0; it0 < ITERS; it0 += CH) {
      float4 res[CH][2];
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int row = rbase + (it0 + q) * RPI;
        const bool ok = p.residual && cok && row < p.m;
#pragma unroll
        for (int h = 0; h < 2; ++h)
          res[q][h] = (ok && (h == 0 || full)) ? *(const float4*)(p.residual + (int64_t)row * p.ldc + col + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int row = rbase + (it0 + q) * RPI;
        const float* src = ep + (rsub + (it0 + q) * RPI) * PITCH + c8;
        const float4 a0 = *(const float4*)src, a1 = *(const float4*)(src + 4);
        if (!cok || row >= p.m) continue;
        float v[8] = {fmaf(a0.x, p.scale, b4[0].x), fmaf(a0.y, p.scale, b4[0].y), fmaf(a0.z, p.scale, b4[0].z), fmaf(a0.w, p.scale, b4[0].w),
                      fmaf(a1.x, p.scale, b4[1].x), fmaf(a1.y, p.scale, b4[1].y), fmaf(a1.z, p.scale, b4[1].z), fmaf(a1.w, p.scale, b4[1].w)};
        if (p.act == GIMS_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        v[0] += res[q][0].x; v[1] += res[q][0].y; v[2] += res[q][0].z; v[3] += res[q][0].w;
        v[4] += res[q][1].x; v[5] += res[q][1].y; v[6] += res[q][1].z; v[7] += res[q][1].w;
        if (p.out_f32) {
          float* o = p.out_f32 + (int64_t)row * p.ldc + col;
          *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
          if (full) *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        uint32_t h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
        if (p.out_bf16) {
          uint16_t* o = p.out_bf16 + (int64_t)row * p.ldc_bf16 + col;
          uint32_t q[4] = {h[0], h[1], h[2], h[3]};
          if (p.flags & GIMS_LINEAR_OUT_F16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = pack_h2_sat(v[2 * e], v[2 * e + 1]);
            if (p.range_stat) {          // range of what the half attention will read (eight columns of ONE 256-column block: col % 8 == 0)
              float m8 = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))),
                               full ? fmaxf(fmaxf(fabsf(v[4]), fabsf(v[5])), fmaxf(fabsf(v[6]), fabsf(v[7]))) : 0.f);
              // fmaxf drops NaN operands: a NaN in what the half attention will read must reach the range row as "not finite" (+inf), or the
              // range guard would pass it (x - x is 0 for finite x, NaN for NaN and inf: one sum finds either)
              const float nf = ((v[0] - v[0]) + (v[1] - v[1])) + ((v[2] - v[2]) + (v[3] - v[3]))
                               + (full ? ((v[4] - v[4]) + (v[5] - v[5])) + ((v[6] - v[6]) + (v[7] - v[7])) : 0.f);
              m8 = nf == 0.f ? m8 : INFINITY;
              const int blk = col >> 8;
              rmax[0] = blk == 0 ? fmaxf(rmax[0], m8) : rmax[0];
              rmax[1] = blk == 1 ? fmaxf(rmax[1], m8) : rmax[1];
              rmax[2] = blk == 2 ? fmaxf(rmax[2], m8) : rmax[2];
            }
          }
          if (full && (p.ldc_bf16 & 7) == 0) *(uint4*)o = make_uint4(q[0], q[1], q[2], q[3]);
          else {
            *(uint2*)o = make_uint2(q[0], q[1]);
            if (full) *(uint2*)(o + 4) = make_uint2(q[2], q[3]);
          }
        }
        if (p.out_hi) {
          uint32_t l[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            l[e] = pack_bf2(v[2 * e] - __uint_as_float(h[e] << 16), v[2 * e + 1] - __uint_as_float(h[e] & 0xffff0000u));
          uint16_t* oh = p.out_hi + (int64_t)row * p.ld_split + scol;
          uint16_t* ol = p.out_lo + (int64_t)row * p.ld_split + scol;
          if (full) { *(uint4*)oh = make_uint4(h[0], h[1], h[2], h[3]); *(uint4*)ol = make_uint4(l[0], l[1], l[2], l[3]); }
          else { *(uint2*)oh = make_uint2(h[0], h[1]); *(uint2*)ol = make_uint2(l[0], l[1]); }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read-back complete before the slice is overwritten
   }
  }
  if ((p.flags & GIMS_LINEAR_OUT_F16) && p.range_stat) {
    // one candidate per wave and block; the atomic only when it beats what is already there (a stale read can only be too small: the atomic then
    // happens needlessly, never the other way round) -- positive floats order like their bit patterns
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float m = wave_max(rmax[b]);
      if (lane == 0 && m > 0.f) {
        unsigned long long* dst = (unsigned long long*)p.range_stat + b;
        if ((unsigned long long)__float_as_uint(m) > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, (unsigned long long)__float_as_uint(m));
      }
    }
  }
}

template <int TM, int TN, int WM, int WN, int S, int HI_ONLY = 0>
__global__ __launch_bounds__(64 * WM * WN) void linear_x3p_kernel(gims_linear_args p) {
  if (p.guard.stat && !attn_guard_fires(p.guard)) return;      // guarded launch (uniform for the grid): nothing to redo
  linear_x3p_tile<TM, TN, WM, WN, S, HI_ONLY>(p, (int)blockIdx.x);
}
// GUARDED launches of the big tile (the device-side redo of attention_precision='auto', include/gims_hip.h gims_attn_guard): a launch that does
// not fire used to dispatch every workgroup of the full grid -- 1536 workgroups with 128 KB of LDS each, six rounds per CU, 7 us -- just to
// have them all return.  Here ONE round of workgroups is dispatched (gridDim.x <= CUs); a launch that fires walks its tiles with that stride
// (same tile -> XCD mapping: the stride is a multiple of 8; same arithmetic per tile: bit-identical to the unguarded launch).
template <int TM, int TN, int WM, int WN, int S, int HI_ONLY = 0>
__global__ __launch_bounds__(64 * WM * WN) void linear_x3p_guarded_kernel(gims_linear_args p, int n_tiles) {
  if (p.guard.stat && !attn_guard_fires(p.guard)) return;
  for (int bid = (int)blockIdx.x; bid < n_tiles; bid += (int)gridDim.x) {
    linear_x3p_tile<TM, TN, WM, WN, S, HI_ONLY>(p, bid);
    __syncthreads();                                            // the next tile's LDS-DMA overwrites the ring / the epilogue slices
  }
}

__global__ void put_linear_args_kernel(gims_linear_args a, gims_linear_args* __restrict__ dst) { *dst = a; }

// ------------------------------------------------------------------------------------------ split kernel
__global__ void split_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ hi,
                                  uint16_t* __restrict__ lo, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float x = src[i];
    const uint16_t h = f2bf(x);
    hi[i] = h;
    lo[i] = f2bf(x - bf2f(h));
  }
}

// f32 [rows][k] -> SPL32 bf16 [rows][2k]
__global__ void split_spl32_kernel(const float* __restrict__ src, int64_t lds, uint16_t* __restrict__ dst, int64_t ldd,
                                   int64_t rows, int k) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = rows * k, stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int64_t r = i / k;
    const int c = (int)(i - r * k);
    const float x = src[r * lds + c];
    const uint16_t h = f2bf(x);
    dst[r * ldd + spl_col(c)] = h;
    dst[r * ldd + spl_col(c) + 32] = f2bf(x - bf2f(h));
  }
}

}  // namespace gims

static int linear_validate(const gims_linear_args* a) {
  using namespace gims;
  GIMS_CHECK_ARG(a != nullptr, "gims_linear: null args");
  GIMS_CHECK_ARG(a->m > 0 && a->n > 0 && a->k > 0, "gims_linear: empty problem m=%d n=%d k=%d", a->m, a->n, a->k);
  GIMS_CHECK_ARG(a->a0 && a->w, "gims_linear: null operand");
  GIMS_CHECK_ARG(a->k0 > 0 && a->k0 <= a->k, "gims_linear: k0=%d out of range", a->k0);
  GIMS_CHECK_ARG(a->k0 == a->k || a->a1 != nullptr, "gims_linear: second A segment missing");
  GIMS_CHECK_ARG(a->out_f32 || a->out_bf16 || a->out_hi, "gims_linear: no output");
  GIMS_CHECK_ARG((a->out_hi == nullptr) == (a->out_lo == nullptr), "gims_linear: out_hi and out_lo come together");
  GIMS_CHECK_ARG(!a->range_stat || (a->a0_lo && (a->flags & GIMS_LINEAR_OUT_F16) && a->out_bf16 && a->n <= 768 && (((uintptr_t)a->range_stat) & 7) == 0),
                 "gims_linear: range_stat goes with pre-split operands, GIMS_LINEAR_OUT_F16, out_bf16, n <= 768, 8-byte aligned");
  GIMS_CHECK_ARG(!a->guard.stat || (a->a0_lo && (a->guard.kind == GIMS_GUARD_PEAKED || a->guard.kind == GIMS_GUARD_RANGE) && a->guard.n_heads > 0 &&
                                    a->guard.n_heads <= 15 && (((uintptr_t)a->guard.stat) & 7) == 0),
                 "gims_linear: a guard goes with pre-split operands, kind GIMS_GUARD_*, 8-byte aligned stat");
  if (a->precision == GIMS_PREC_BF16X6) {   // SPL3 operands, batched launches only: C = scale * A W^T, f32 out
    GIMS_CHECK_ARG((a->k % 32) == 0 && a->k0 == a->k, "gims_linear(bf16x6): K=%d must be a multiple of 32, one A segment", a->k);
    GIMS_CHECK_ARG(a->lda0 >= 3 * (int64_t)a->k && a->ldw >= 3 * (int64_t)a->k && (a->lda0 % 8) == 0 && (a->ldw % 8) == 0,
                   "gims_linear(bf16x6): SPL3 operands have row pitch >= 3*K, a multiple of 8 elements");
    GIMS_CHECK_ARG((((uintptr_t)a->a0 | (uintptr_t)a->w | (uintptr_t)a->out_f32) & 15) == 0 && a->out_f32 && (a->ldc % 4) == 0,
                   "gims_linear(bf16x6): operands / output must be 16-byte aligned, f32 output with ldc %% 4 == 0");
    GIMS_CHECK_ARG(!a->bias && !a->residual && !a->out_bf16 && !a->out_hi && a->act == GIMS_ACT_NONE,
                   "gims_linear(bf16x6): plain scaled product only");
    return GIMS_OK;
  }
  if (a->flags & GIMS_LINEAR_CONV3) {
    GIMS_CHECK_ARG(a->a0_lo && a->a1 && a->k0 == a->k && (a->k % 9) == 0 && ((a->k / 9) % 32) == 0 && a->conv_h > 0 && a->conv_w > 0 &&
                       (a->conv_stride == 1 || a->conv_stride == 2) &&
                       (a->m % (((a->conv_h - 1) / a->conv_stride + 1) * ((a->conv_w - 1) / a->conv_stride + 1))) == 0,
                   "gims_linear(conv3): pre-split NHWC input with C %% 32 == 0, k = 9 C, a1 = 128 zero bytes, m = patches * Ho * Wo");
  }
  if (a->a0_lo) {   // pre-split activations: bf16 hi/lo planes, LDS-DMA kernel
    GIMS_CHECK_ARG(a->precision == GIMS_PREC_BF16X3 && a->w_lo, "gims_linear: pre-split A needs GIMS_PREC_BF16X3 and w_lo");
    GIMS_CHECK_ARG(a->k0 == a->k || a->a1_lo, "gims_linear: second A segment needs its lo plane");
    GIMS_CHECK_ARG((a->k % 32) == 0 && (a->k0 % 32) == 0, "gims_linear(pre-split): K=%d k0=%d must be multiples of 32", a->k, a->k0);
    GIMS_CHECK_ARG((a->lda0 % 64) == 0 && (a->lda1 % 64) == 0 && (a->ldw % 64) == 0 && a->lda0 >= 2 * ((a->flags & GIMS_LINEAR_CONV3) ? a->k / 9 : a->k0) && a->ldw >= 2 * a->k,
                   "gims_linear(pre-split): SPL32 operands have row pitch >= 2*K, a multiple of 64 elements");
    GIMS_CHECK_ARG((((uintptr_t)a->a0 | (uintptr_t)a->w | (uintptr_t)a->a1) & 127) == 0, "gims_linear(pre-split): SPL32 operands must be 128-byte aligned");
    GIMS_CHECK_ARG(a->lda0 < (1 << 22) && a->lda1 < (1 << 22) && a->ldw < (1 << 22), "gims_linear(pre-split): row pitch too large (32-bit tile-relative offsets)");
    GIMS_CHECK_ARG((a->n % 4) == 0 && (a->ldc % 4) == 0 && (a->ldc_bf16 % 4) == 0 && (a->ld_split % 8) == 0,
                   "gims_linear(pre-split): n, ldc and ldc_bf16 must be multiples of 4, ld_split of 8");
    GIMS_CHECK_ARG((((uintptr_t)a->out_f32 | (uintptr_t)a->out_hi | (uintptr_t)a->residual | (uintptr_t)a->bias) & 15) == 0 &&
                   ((uintptr_t)a->out_bf16 & 7) == 0 && (((uintptr_t)a->out_bf16 & 15) == 0 || (a->ldc_bf16 & 7) != 0),
                   "gims_linear(pre-split): bias, residual and outputs must be 16-byte aligned");
    return GIMS_OK;
  }
  GIMS_CHECK_ARG(!a->residual || a->out_f32, "gims_linear: residual needs an f32 output (shared ldc)");
  GIMS_CHECK_ARG((a->lda0 % 4) == 0 && (a->lda1 % 4) == 0, "gims_linear: lda must be a multiple of 4");
  if (a->precision == GIMS_PREC_F32) {
    GIMS_CHECK_ARG((a->k % F32_BK) == 0 && (a->k0 % F32_BK) == 0, "gims_linear(f32): K=%d k0=%d must be multiples of %d", a->k, a->k0, F32_BK);
    GIMS_CHECK_ARG((a->ldw % 4) == 0, "gims_linear(f32): ldw must be a multiple of 4");
  } else if (a->precision == GIMS_PREC_BF16X3) {
    GIMS_CHECK_ARG((a->k % X3_BK) == 0 && (a->k0 % X3_BK) == 0, "gims_linear(bf16x3): K=%d k0=%d must be multiples of %d", a->k, a->k0, X3_BK);
    GIMS_CHECK_ARG(a->w_lo != nullptr && (a->ldw % 8) == 0, "gims_linear(bf16x3): needs w_lo and ldw %% 8 == 0");
  } else {
    GIMS_CHECK_ARG(false, "gims_linear: unknown precision %d", a->precision);
  }
  return GIMS_OK;
}

static int x3_attr() {
  using namespace gims;
  const int lds = 4 * X3_PLANE * (int)sizeof(uint16_t);
  GIMS_LDS_ATTR((const void*)linear_bf16x3_kernel, lds);
  GIMS_LDS_ATTR((const void*)linear_bf16x3_batch_kernel, lds);
  return GIMS_OK;
}

extern "C" int gims_linear(const gims_linear_args* a, void* stream) {
  using namespace gims;
  int rc = linear_validate(a);
  if (rc != GIMS_OK) return rc;
  GIMS_CHECK_ARG(a->precision != GIMS_PREC_BF16X6, "gims_linear: GIMS_PREC_BF16X6 runs through gims_linear_put_many + gims_linear_batch");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(cdiv(a->n, BN), cdiv(a->m, BM));
  if (a->a0_lo) {
    // tile geometry: 256x256 (half the operand bytes per MFMA) when that still gives ~one workgroup per CU,
    // else 128x128.  GIMS_X3P_TILE=128|256 forces one (A/B).
    using TS = X3P<128, 128, 2, 2, 2>;
    using TL = X3P<256, 256, 4, 2, 2>;
    static const int force = [] { const char* e = getenv("GIMS_X3P_TILE"); return e ? atoi(e) : 0; }();
    // one-pass (Q/K/V) GEMMs: 0 = 256 x 256 tiles like the others; 2 / 3 = 256 x 128 tiles with that many ring stages (default 3)
    static const int qkv_tile = [] { const char* e = getenv("GIMS_X3P_QKV"); return e ? atoi(e) : 3; }();
    static const int small_tiles = [] { const char* e = getenv("GIMS_X3P_SMALL"); return e ? atoi(e) : 128; }();      // 128 x 128 tiles at or below which a launch takes 64 x 64 ones (GIMS_X3P_SMALL=0: never)
    const int big_blocks = cdiv(a->m, 256) * cdiv(a->n, 256);
    // the 256-wide tile only when it is not half empty (n = 64 / 128 layers of the keypoint encoder and GraphSAGE)
    const bool big = force == 256 || (force != 128 && big_blocks >= 192 && (a->n % 256 == 0 || a->n > 512));
    if (big && (a->flags & GIMS_LINEAR_HI_ONLY) && qkv_tile > 0) {
      // one-pass GEMMs with a short K (the Q/K/V projection: K = 256, 8 stages) are all prologue and epilogue: 256 x 128
      // tiles with 64 accumulator registers per wave let TWO workgroups share a CU, one's epilogue under the other's loads
      using TQ2 = X3P<256, 128, 4, 2, 2, true>;
      using TQ3 = X3P<256, 128, 4, 2, 3, true>;
      GIMS_LDS_ATTR((const void*)linear_x3p_kernel<256, 128, 4, 2, 2, 1>, (int)TQ2::LDS_BYTES);
      GIMS_LDS_ATTR((const void*)linear_x3p_kernel<256, 128, 4, 2, 3, 1>, (int)TQ3::LDS_BYTES);
      const dim3 g(8 * cdiv(cdiv(a->m, 256), 8) * cdiv(a->n, 128));
      if (qkv_tile == 2) { constexpr size_t lds = TQ2::LDS_BYTES; hipLaunchKernelGGL((linear_x3p_kernel<256, 128, 4, 2, 2, 1>), g, dim3(512), lds, s, *a); }
      else { constexpr size_t lds = TQ3::LDS_BYTES; hipLaunchKernelGGL((linear_x3p_kernel<256, 128, 4, 2, 3, 1>), g, dim3(512), lds, s, *a); }
    } else if (big) {
      using TLH = X3P<256, 256, 4, 2, 4, true>;
      constexpr size_t lds = TL::LDS_BYTES, lds_h = TLH::LDS_BYTES;
      GIMS_LDS_ATTR((const void*)linear_x3p_kernel<256, 256, 4, 2, 4, 1>, (int)lds_h);
      GIMS_LDS_ATTR((const void*)linear_x3p_kernel<256, 256, 4, 2, 2, 2>, (int)lds);
      GIMS_LDS_ATTR((const void*)linear_x3p_kernel<256, 256, 4, 2, 2>, (int)lds);
      const dim3 g(8 * cdiv(cdiv(a->m, 256), 8) * cdiv(a->n, 256));
      if (a->flags & GIMS_LINEAR_HI_ONLY) hipLaunchKernelGGL((linear_x3p_kernel<256, 256, 4, 2, 4, 1>), g, dim3(512), lds_h, s, *a);
      else if (a->guard.stat && (int)g.x > device_cus() && guard_walk_enabled()) {      // guarded: one round of workgroups (see linear_x3p_guarded_kernel)
        GIMS_LDS_ATTR((const void*)linear_x3p_guarded_kernel<256, 256, 4, 2, 2>, (int)lds);
        hipLaunchKernelGGL((linear_x3p_guarded_kernel<256, 256, 4, 2, 2>), dim3(device_cus() & ~7), dim3(512), lds, s, *a, (int)g.x);
      } else hipLaunchKernelGGL((linear_x3p_kernel<256, 256, 4, 2, 2>), g, dim3(512), lds, s, *a);
    } else if (a->flags & GIMS_LINEAR_CONV3) {
      using T32 = X3P<128, 32, 4, 1, 2>;
      using T64 = X3P<128, 64, 2, 2, 2>;
      GIMS_LDS_ATTR((const void*)linear_x3p_kernel<128, 32, 4, 1, 2, 4>, (int)T32::LDS_BYTES);
      GIMS_LDS_ATTR((const void*)linear_x3p_kernel<128, 64, 2, 2, 2, 4>, (int)T64::LDS_BYTES);
      GIMS_LDS_ATTR((const void*)linear_x3p_kernel<128, 128, 2, 2, 2, 4>, (int)TS::LDS_BYTES);
      const int mt = 8 * cdiv(cdiv(a->m, 128), 8);
      if (a->n <= 32) { constexpr size_t lds = T32::LDS_BYTES; hipLaunchKernelGGL((linear_x3p_kernel<128, 32, 4, 1, 2, 4>), dim3(mt * cdiv(a->n, 32)), dim3(256), lds, s, *a); }
      else if (a->n <= 64) { constexpr size_t lds = T64::LDS_BYTES; hipLaunchKernelGGL((linear_x3p_kernel<128, 64, 2, 2, 2, 4>), dim3(mt * cdiv(a->n, 64)), dim3(256), lds, s, *a); }
      else { constexpr size_t lds = TS::LDS_BYTES; hipLaunchKernelGGL((linear_x3p_kernel<128, 128, 2, 2, 2, 4>), dim3(mt * cdiv(a->n, 128)), dim3(256), lds, s, *a); }
    } else if (a->n <= 64 && !(a->flags & GIMS_LINEAR_HI_ONLY) && force == 0) {
      // narrow outputs (the 32- and 64-channel convolutions of the descriptor network, millions of rows): 128 x 32 / 128 x 64
      // tiles instead of wasting three quarters / half of a 128-wide one
      using T32 = X3P<128, 32, 4, 1, 2>;
      using T64 = X3P<128, 64, 2, 2, 2>;
      GIMS_LDS_ATTR((const void*)linear_x3p_kernel<128, 32, 4, 1, 2>, (int)T32::LDS_BYTES);
      GIMS_LDS_ATTR((const void*)linear_x3p_kernel<128, 64, 2, 2, 2>, (int)T64::LDS_BYTES);
      if (a->n <= 32) {
        constexpr size_t lds = T32::LDS_BYTES;
        hipLaunchKernelGGL((linear_x3p_kernel<128, 32, 4, 1, 2>), dim3(8 * cdiv(cdiv(a->m, 128), 8) * cdiv(a->n, 32)), dim3(256), lds, s, *a);
      } else {
        constexpr size_t lds = T64::LDS_BYTES;
        hipLaunchKernelGGL((linear_x3p_kernel<128, 64, 2, 2, 2>), dim3(8 * cdiv(cdiv(a->m, 128), 8) * cdiv(a->n, 64)), dim3(256), lds, s, *a);
      }
    } else if (force == 64 || (force == 0 && cdiv(a->m, 128) * cdiv(a->n, 128) <= small_tiles)) {
      // launches that leave most of the chip idle at 128 x 128 (one pair through forward(): 32 ... 96 tiles at 2 x 1024 keypoints): 64 x 64 tiles
      // on four waves -- four times the workgroups, the same K order per output element (bit-identical), and what a launch costs there is
      // the latency of its K loop, not its matrix work
      using T6 = X3P<64, 64, 2, 2, 3>;
      using T6H = X3P<64, 64, 2, 2, 4, true>;
      GIMS_LDS_ATTR((const void*)linear_x3p_kernel<64, 64, 2, 2, 3>, (int)T6::LDS_BYTES);
      GIMS_LDS_ATTR((const void*)linear_x3p_kernel<64, 64, 2, 2, 4, 1>, (int)T6H::LDS_BYTES);
      const dim3 g(8 * cdiv(cdiv(a->m, 64), 8) * cdiv(a->n, 64));
      if (a->flags & GIMS_LINEAR_HI_ONLY) { constexpr size_t lds = T6H::LDS_BYTES; hipLaunchKernelGGL((linear_x3p_kernel<64, 64, 2, 2, 4, 1>), g, dim3(256), lds, s, *a); }
      else { constexpr size_t lds = T6::LDS_BYTES; hipLaunchKernelGGL((linear_x3p_kernel<64, 64, 2, 2, 3>), g, dim3(256), lds, s, *a); }
    } else {
      using TSH = X3P<128, 128, 2, 2, 4, true>;
      constexpr size_t lds = TS::LDS_BYTES, lds_h = TSH::LDS_BYTES;
      GIMS_LDS_ATTR((const void*)linear_x3p_kernel<128, 128, 2, 2, 4, 1>, (int)lds_h);
      GIMS_LDS_ATTR((const void*)linear_x3p_kernel<128, 128, 2, 2, 2, 2>, (int)lds);
      GIMS_LDS_ATTR((const void*)linear_x3p_kernel<128, 128, 2, 2, 2>, (int)lds);
      const dim3 g(8 * cdiv(cdiv(a->m, 128), 8) * cdiv(a->n, 128));
      if (a->flags & GIMS_LINEAR_HI_ONLY) hipLaunchKernelGGL((linear_x3p_kernel<128, 128, 2, 2, 4, 1>), g, dim3(256), lds_h, s, *a);
      else if (force == 128 || cdiv(a->m, 128) * cdiv(a->n, 128) > 256)      // more than one tile per CU (or GIMS_X3P_TILE=128): the 4-wave tile, two workgroups per CU
        hipLaunchKernelGGL((linear_x3p_kernel<128, 128, 2, 2, 2>), g, dim3(256), lds, s, *a);
      else {
        // small launches are latency-bound (one tile per CU, ~1 us per K step at one wave per SIMD): the same 128 x 128 tile on
        // EIGHT waves (two per SIMD, 64 x 32 ... per wave) hides the LDS and MFMA-chain latencies: 23 -> 18 us at 8192 rows,
        // 21 -> 15 us at 2048 rows (tools/gemm_probe.py); same K order per output element: bit-identical results
        GIMS_LDS_ATTR((const void*)linear_x3p_kernel<128, 128, 4, 2, 2>, (int)X3P<128, 128, 4, 2, 2>::LDS_BYTES);
        constexpr size_t lds8 = X3P<128, 128, 4, 2, 2>::LDS_BYTES;
        hipLaunchKernelGGL((linear_x3p_kernel<128, 128, 4, 2, 2>), g, dim3(512), lds8, s, *a);
      }
    }
  } else if (a->precision == GIMS_PREC_F32) {
    hipLaunchKernelGGL(linear_f32_kernel, grid, dim3(256), 0, s, *a);
  } else {
    if ((rc = x3_attr()) != GIMS_OK) return rc;
    hipLaunchKernelGGL(linear_bf16x3_kernel, grid, dim3(256), 4 * X3_PLANE * sizeof(uint16_t), s, *a);
  }
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_linear_put(const gims_linear_args* a, gims_linear_args* dev_dst, void* stream) {
  using namespace gims;
  int rc = linear_validate(a);
  if (rc != GIMS_OK) return rc;
  GIMS_CHECK_ARG(dev_dst != nullptr, "gims_linear_put: null destination");
  hipLaunchKernelGGL(put_linear_args_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, *a, dev_dst);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_linear_put_many(const gims_linear_args* h_args, int32_t count, gims_linear_args* dev_dst, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(h_args && count > 0 && dev_dst, "gims_linear_put_many: bad arguments");
  for (int i = 0; i < count; ++i) {
    const int rc = linear_validate(h_args + i);
    if (rc != GIMS_OK) return rc;
    GIMS_CHECK_ARG(h_args[i].a0_lo == nullptr, "gims_linear_put_many: pre-split operands are not available in batches");
  }
  return upload_table(h_args, sizeof(gims_linear_args) * (size_t)count, dev_dst, (hipStream_t)stream);
}

extern "C" int gims_linear_batch(const gims_linear_args* dev_args, int32_t count, int32_t max_m, int32_t max_n,
                                 int32_t precision, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(dev_args && count > 0 && max_m > 0 && max_n > 0, "gims_linear_batch: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(cdiv(max_n, BN), cdiv(max_m, BM), count);
  if (precision == GIMS_PREC_BF16X6) {
    return linear_x6_batch_launch(dev_args, count, max_m, max_n, s);
  } else if (precision == GIMS_PREC_F32) {
    hipLaunchKernelGGL(linear_f32_batch_kernel, grid, dim3(256), 0, s, dev_args);
  } else if (precision == GIMS_PREC_BF16X3) {
    int rc = x3_attr();
    if (rc != GIMS_OK) return rc;
    hipLaunchKernelGGL(linear_bf16x3_batch_kernel, grid, dim3(256), 4 * X3_PLANE * sizeof(uint16_t), s, dev_args);
  } else {
    GIMS_CHECK_ARG(false, "gims_linear_batch: unknown precision %d", precision);
  }
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_split_spl32(const float* src, int64_t lds, uint16_t* dst, int64_t ldd, int64_t rows, int32_t k, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(src && dst && rows >= 0 && k > 0 && (k % 32) == 0 && ldd >= 2 * (int64_t)k, "gims_split_spl32: bad arguments");
  if (rows == 0) return GIMS_OK;
  int64_t blocks = (rows * k + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_spl32_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, src, lds, dst, ldd, rows, k);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_split_bf16(const float* src, uint16_t* hi, uint16_t* lo, int64_t n, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(src && hi && lo && n >= 0, "gims_split_bf16: bad args");
  if (n == 0) return GIMS_OK;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, hi, lo, n);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

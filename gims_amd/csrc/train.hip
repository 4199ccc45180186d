// Training-step kernels (forward_train + backward, models/gmatcher.py:309-386, train.py:136-137).
//
// The training step runs in f32-class arithmetic end to end, on ROW-MAJOR f32 activations [keypoint rows][channels]:
//   * gemm_split_kernel   -- one general batched GEMM  C = alpha * op(A) op(B)^T + beta * C (+ bias, + residual, ReLU)
//                            for every product of the step (linear layers forward / input gradient / weight gradient, the
//                            attention products Q K^T, P V and their five gradients, the score matrix and its gradients).
//                            f32 operands are split into bf16 hi + lo on the way into LDS and multiplied with three bf16 MFMA
//                            passes (hi*hi + hi*lo + lo*hi, v_mfma_f32_32x32x16_bf16): ~2^-17 relative, the accuracy class of the
//                            reference's f32 matmuls.  Either operand may be stored transposed ([k][rows]).
//   * BatchNorm1d in train() mode (gmatcher.py:17-22 under nn.Module.train()): batch statistics per call = per SIDE of the
//                            batch (the reference calls every layer once for image 0 and once for image 1), running
//                            statistics updated in call order with the unbiased variance, and its backward pass.
//   * softmax over attention rows and its backward, column sums (bias gradients), the transposed mean aggregation of
//                            GraphSAGE, a strided 3-D copy (head interleave of gmatcher.py:108-113 <-> contiguous heads).
// All reductions have a fixed order: a training step is bitwise reproducible.
#include <type_traits>

#include "common.h"

namespace gims {

// ------------------------------------------------------------------------------------------------ general GEMM, bf16x3
// LDS tile: NS planes (bf16 hi, mid[, lo] of the f32 value) of [rows][32 k], 64 bytes per row.  Bank rules on gfx950 (micro-architecture
// guide, LDS table): ds_read_b128 is served in four fixed 16-lane groups -- lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the
// same + 32 -- over 64 banks (a 256-byte row); ds_write_b64 in contiguous 16-lane groups over 32 banks.  The layout satisfies all three
// access patterns of the kernel:
//   * fragment reads (lane = row): a group holds the row quartets {0, 3, 5, 6} (or {1, 2, 4, 7}) of its 32 rows; a quartet fills the four
//     64-byte segments of a bank row, so the 16-byte chunk position must differ between the quartets of a group: chunk ^ ((row >> 3) & 3);
//   * stores of the row-major loader (8 lanes per row): two neighbouring rows per group, one per 64-byte half -- any layout works;
//   * stores of the TRANSPOSED loader: a group writes ONE chunk of the rows 4 l + r of eight consecutive quartets l, both 8-byte halves
//     (its lane pairs hold k and k + 4): the rows of even and odd quartets must sit in different 64-byte halves -> the physical row swaps
//     neighbours in odd quartets (row ^ ((row >> 2) & 1)), and consecutive quartet pairs take the four chunk positions.
// Measured before (swizzle (row >> 1) & 3 only): SQ_LDS_BANK_CONFLICT 39-49 % of the LDS cycles of the transposed-operand products,
// 22-30 % of the others.
__device__ __forceinline__ int tile_off(int row, int chunk) { return (row ^ ((row >> 2) & 1)) * 32 + ((chunk ^ ((row >> 3) & 3)) << 3); }

// R rows x 32 k of an operand into registers.  T = false: stored [rows][k] (k contiguous); T = true: stored [k][rows].
// The loads are UNCONDITIONAL (indices clamped into the operand) and nothing consumes them here: masking of the ragged edges
// happens in tile_store, two compute phases later -- a select right behind a load would make every load wait for its own data
// and serialise the eight round trips of a tile.
template <bool T, int R, bool VEC>
__device__ __forceinline__ void tile_load(f32x4 (&v)[4], const float* __restrict__ base, int64_t ld, int row0, int nrows, int k0, int K, int t) {
  if constexpr (!T) {
    const int k = k0 + (t & 7) * 4;
#pragma unroll
    for (int i = 0; i < R / 32; ++i) {
      const int gr = min(row0 + (t >> 3) + 32 * i, nrows - 1);
      const float* p = base + (int64_t)gr * ld;
      if constexpr (VEC) {
        v[i] = *(const f32x4*)(p + min(k, (K - 1) & ~3));
      } else {
        v[i] = f32x4{p[min(k, K - 1)], p[min(k + 1, K - 1)], p[min(k + 2, K - 1)], p[min(k + 3, K - 1)]};
      }
    }
  } else {
    // lane pairs hold k-blocks kb and kb + 1 of the same four rows (see tile_off); both still read 512 contiguous bytes per k-row
    const int rq = row0 + ((t >> 1) % (R / 4)) * 4, kb = min((t & 1) + 2 * (t / (R / 2)), 7);       // (threads past the tile repeat its last block: no branch)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* p = base + (int64_t)min(k0 + kb * 4 + j, K - 1) * ld;
      if constexpr (VEC) {
        v[j] = *(const f32x4*)(p + min(rq, (nrows - 1) & ~3));
      } else {
        v[j] = f32x4{p[min(rq, nrows - 1)], p[min(rq + 1, nrows - 1)], p[min(rq + 2, nrows - 1)], p[min(rq + 3, nrows - 1)]};
      }
    }
  }
}

template <int NS, int R>
__device__ __forceinline__ void item_store(uint16_t* __restrict__ tile, int row, int k4, f32x4 x) {
  const int o = tile_off(row, k4 >> 3) + (k4 & 4);
#pragma unroll
  for (int pl = 0; pl < NS; ++pl) {
    const uint32_t h01 = pack_bf2(x[0], x[1]), h23 = pack_bf2(x[2], x[3]);
    *(uint2*)(tile + pl * (R * 32) + o) = make_uint2(h01, h23);
    x[0] -= __uint_as_float(h01 << 16);
    x[1] -= __uint_as_float(h01 & 0xffff0000u);
    x[2] -= __uint_as_float(h23 << 16);
    x[3] -= __uint_as_float(h23 & 0xffff0000u);
  }
}

// registers of tile_load -> LDS planes; zeroes what lies outside [row0, nrows) x [k0, K)
template <bool T, int R, int NS>
__device__ __forceinline__ void tile_store(const f32x4 (&v)[4], uint16_t* __restrict__ tile, int row0, int nrows, int k0, int K, int t) {
  if constexpr (!T) {
    const int k4 = (t & 7) * 4, k = k0 + k4;
#pragma unroll
    for (int i = 0; i < R / 32; ++i) {
      const int row = (t >> 3) + 32 * i;
      const bool rv = row0 + row < nrows;
      f32x4 x = v[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = (rv && k + j < K) ? x[j] : 0.f;
      item_store<NS, R>(tile, row, k4, x);
    }
  } else {
    const int rq = ((t >> 1) % (R / 4)) * 4, kb = (t & 1) + 2 * (t / (R / 2));
    if (kb < 8) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool rv = row0 + rq + r < nrows;
        f32x4 x;
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = (rv && k0 + kb * 4 + j < K) ? v[j][r] : 0.f;
        item_store<NS, R>(tile, rq + r, kb * 4, x);
      }
    }
  }
}

// NS = 2: split-bf16x3 (hi*hi + hi*mid + mid*hi, 16 mantissa bits per operand); NS = 3: split-bf16x6 (+ mid*mid + hi*lo + lo*hi,
// 24 bits: the f32 class)
__device__ unsigned long long g_gemm_prof[8];     // GIMS_GEMM_PROF=1: cycle stamps of one workgroup (diagnostics)

// VEC: both operands 16-byte aligned with pitches that are multiples of 4 (straight-line loads; the compiler then counts the
// loads in flight and waits only for the register set it is about to use)
template <bool TA, bool TB, int BM, int BN, int NS, bool VEC>
__global__ __launch_bounds__(256) void gemm_split_kernel(gims_gemm g) {
  // wave grid 2 x 2 (4 x 1 for 64-column tiles); a wave owns MT x 2 MFMA tiles of 32 x 32
  static_assert((BM == 128 || BM == 64) && (BN == 128 || BN == 64) && !(BM == 64 && BN == 64), "tile geometry");
  constexpr int MT = (BM == 128 && BN == 128) ? 2 : 1;
  // ONE LDS stage (48 KB at 128 x 128, three planes: two or three workgroups per CU hide each other's barriers and the
  // epilogue) fed from TWO register sets in flight: tile kt + 2 is requested right after tile kt went to LDS, so a global load
  // has two compute phases to land
  __shared__ __attribute__((aligned(16))) uint16_t smem[NS * (BM + BN) * 32];
  uint16_t* const As = smem;
  uint16_t* const Bs = smem + NS * BM * 32;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int nsplit = g.splits > 1 ? g.splits : 1;
  const int z = blockIdx.z / nsplit, split = blockIdx.z - z * nsplit;
  // split-K: this workgroup covers k in [kbeg, kend) and leaves its raw partial sums in the workspace (fixed-order reduction
  // and the epilogue follow in splitk_reduce_kernel -- folding inside this kernel by the last split to arrive was tried: the
  // partial sums then need device-scope stores to cross the XCDs' L2s, which made the weight-gradient GEMMs 3x slower)
  const int kchunk = ((g.k + nsplit - 1) / nsplit + 31) & ~31;
  const int kbeg = split * kchunk, kend = min(g.k, kbeg + kchunk);
  const int wm = BN == 64 ? wave * 32 : (wave >> 1) * (32 * MT), wn = BN == 64 ? 0 : (wave & 1) * 64;
  const float* __restrict__ A = g.a + (int64_t)z * g.sa;
  const float* __restrict__ B = g.b + (int64_t)z * g.sb;
  const bool prof = (g.flags & 4) && blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && blockIdx.z == 0 && t == 0;
  if (prof) { g_gemm_prof[0] = __builtin_readcyclecounter(); g_gemm_prof[4] = g_gemm_prof[5] = g_gemm_prof[6] = g_gemm_prof[7] = 0; }

  f32x16 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[2][4], rb[2][4];
  const int nk = kend > kbeg ? (kend - kbeg + 31) / 32 : 0;
  if (nk > 0) {
    tile_load<TA, BM, VEC>(ra[0], A, g.lda, m0, g.m, kbeg, kend, t);
    tile_load<TB, BN, VEC>(rb[0], B, g.ldb, n0, g.n, kbeg, kend, t);
    tile_load<TA, BM, VEC>(ra[1], A, g.lda, m0, g.m, kbeg + (nk > 1 ? 32 : 0), kend, t);
    tile_load<TB, BN, VEC>(rb[1], B, g.ldb, n0, g.n, kbeg + (nk > 1 ? 32 : 0), kend, t);
    tile_store<TA, BM, NS>(ra[0], As, m0, g.m, kbeg, kend, t);
    tile_store<TB, BN, NS>(rb[0], Bs, n0, g.n, kbeg, kend, t);
  }
  if (prof) g_gemm_prof[1] = __builtin_readcyclecounter();
  // step kt (tile kt is in LDS, tile kt + 1 in flight in the other register set): request tile kt + 2 into the set that was just
  // stored, multiply, then move tile kt + 1 to LDS.  The loop header sits right before an ISSUE point on purpose: the compiler's
  // load counter is imprecise across the back edge, and the first wait behind it must not be the one that decides how far
  // ahead the loads run (with the store first, it waited for the newest loads as well: prefetch distance one instead of two).
  auto step = [&](auto set_c, int kt) {
    constexpr int SET = decltype(set_c)::value;
    unsigned long long c0 = 0;
    if (prof) c0 = __builtin_readcyclecounter();
    {   // past the end: the last tile again -- unconditional, so the loop body stays straight-line code
      const int kn = kbeg + min(kt + 2, nk - 1) * 32;
      tile_load<TA, BM, VEC>(ra[SET], A, g.lda, m0, g.m, kn, kend, t);
      tile_load<TB, BN, VEC>(rb[SET], B, g.ldb, n0, g.n, kn, kend, t);
    }
    __syncthreads();
    if (prof) { const unsigned long long c1 = __builtin_readcyclecounter(); g_gemm_prof[4] += c1 - c0; c0 = c1; }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[NS][MT], bfr[NS][2];
#pragma unroll
      for (int pl = 0; pl < NS; ++pl) {
#pragma unroll
        for (int i = 0; i < MT; ++i) af[pl][i] = *(const bf16x8*)(As + pl * (BM * 32) + tile_off(wm + i * 32 + li, ks * 2 + lh));
#pragma unroll
        for (int j = 0; j < 2; ++j) bfr[pl][j] = *(const bf16x8*)(Bs + pl * (BN * 32) + tile_off(wn + j * 32 + li, ks * 2 + lh));
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // small terms first
          if constexpr (NS == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bfr[1][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[2][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bfr[0][j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bfr[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[0][j], acc[i][j], 0, 0, 0);
        }
    }
    if (prof) { const unsigned long long c1 = __builtin_readcyclecounter(); g_gemm_prof[5] += c1 - c0; c0 = c1; }
    __syncthreads();
    if (prof) { const unsigned long long c1 = __builtin_readcyclecounter(); g_gemm_prof[6] += c1 - c0; c0 = c1; }
    tile_store<TA, BM, NS>(ra[SET ^ 1], As, m0, g.m, kbeg + (kt + 1) * 32, kend, t);
    tile_store<TB, BN, NS>(rb[SET ^ 1], Bs, n0, g.n, kbeg + (kt + 1) * 32, kend, t);
    if (prof) { const unsigned long long c1 = __builtin_readcyclecounter(); g_gemm_prof[7] += c1 - c0; }
  };
  // (eight steps per trip -- no back edge for k <= 256 -- were tried: exact waits everywhere, but the hoisted addressing of eight
  // steps spills; two steps per trip it is)
  for (int kt = 0; kt < nk; kt += 2) {
    step(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nk) step(std::integral_constant<int, 1>{}, kt + 1);
  }
  if (prof) g_gemm_prof[2] = __builtin_readcyclecounter();

  // ---- epilogue.  The accumulator layout (a lane holds ONE column of 16 rows) turns a direct store into 64 four-byte stores per
  // thread, two 128-byte row pieces per instruction: measured, that was 12 k of the 37 k cycles of a 64 x 128 linear tile and 24 k of
  // the 37 k of a 128 x 128, k = 64 attention product.  So the tile goes through LDS 32 rows at a time (the operand stages are
  // dead) and leaves as 16-byte stores, a wave covering whole rows; bias / residual / beta C are read the same way.
  constexpr int CP = BN + 4;                                 // LDS pitch of a staged row (floats)
  static_assert(32 * CP * 4 <= NS * (BM + BN) * 64, "a 32-row slice of the tile fits the operand stages");
  float* const Cs = (float*)smem;
  const bool split_out = nsplit > 1;
  float* __restrict__ Cz = split_out ? g.work + ((int64_t)split * g.batch + z) * (int64_t)g.m * g.n : g.c + (int64_t)z * g.sc;
  const int64_t ldc = split_out ? g.n : g.ldc;
  const float* __restrict__ Rz = (!split_out && g.residual) ? g.residual + (int64_t)z * g.sr : nullptr;
  const float* __restrict__ bias = split_out ? nullptr : g.bias;
  const float al = split_out ? 1.f : g.alpha, be = split_out ? 0.f : g.beta;
  const int act = split_out ? GIMS_ACT_NONE : g.act;
  const bool vecc = split_out ? (g.flags & 16) != 0 : (g.flags & 8) != 0;       // 16-byte aligned rows of C (and of the residual)
  constexpr int TPR = BN / 4, RPI = 256 / TPR;               // threads per row, rows per iteration of the store loop
  const int cq = (t % TPR) * 4, rr = t / TPR;
  const int n = n0 + cq;
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (bias)
#pragma unroll
    for (int e = 0; e < 4; ++e) b4[e] = n + e < g.n ? bias[n + e] : 0.f;
#pragma unroll
  for (int pass = 0; pass < BM / 32; ++pass) {
    __syncthreads();                                          // the operand stages / the previous slice are no longer read
    {
      const int own_i = pass % MT;
      const bool mine = BN == 64 ? wave == pass : (wave >> 1) == pass / MT;
      if (mine) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          if (i == own_i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) Cs[((r & 3) + 8 * (r >> 2) + 4 * lh) * CP + wn + j * 32 + li] = acc[i][j][r];
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) {
      const int row = rr + it * RPI, m = m0 + pass * 32 + row;
      if (m >= g.m || n >= g.n) continue;
      f32x4 v = *(const f32x4*)(Cs + row * CP + cq);
      float* cp = Cz + (int64_t)m * ldc + n;
      if (vecc && n + 3 < g.n) {
        v = v * al + b4;
        if (Rz) v += *(const f32x4*)(Rz + (int64_t)m * g.ldr + n);
        if (be != 0.f) v += *(const f32x4*)cp * be;
        if (act == GIMS_ACT_RELU) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
        *(f32x4*)cp = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < g.n) {
            float x = al * v[e] + b4[e];
            if (Rz) x += Rz[(int64_t)m * g.ldr + n + e];
            if (be != 0.f) x += be * cp[e];
            if (act == GIMS_ACT_RELU) x = fmaxf(x, 0.f);
            cp[e] = x;
          }
      }
    }
  }
  if (prof) g_gemm_prof[3] = __builtin_readcyclecounter();
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(gims_gemm g) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, per = (int64_t)g.m * g.n;
  if (idx >= per * g.batch) return;
  const int z = (int)(idx / per);
  const int64_t e = idx - (int64_t)z * per;
  const int m = (int)(e / g.n), n = (int)(e - (int64_t)m * g.n);
  float acc = 0.f;
  for (int sp = 0; sp < g.splits; ++sp) acc += g.work[((int64_t)sp * g.batch + z) * per + e];
  float v = g.alpha * acc + (g.bias ? g.bias[n] : 0.f);
  if (g.residual) v += g.residual[(int64_t)z * g.sr + (int64_t)m * g.ldr + n];
  float* cp = g.c + (int64_t)z * g.sc + (int64_t)m * g.ldc + n;
  if (g.beta != 0.f) v += g.beta * *cp;
  if (g.act == GIMS_ACT_RELU) v = fmaxf(v, 0.f);
  *cp = v;
}

template <bool TA, bool TB, int NS, bool VEC>
static void gemm_launch_ns(const gims_gemm& g, hipStream_t s) {
  const int sp = g.splits > 1 ? g.splits : 1;
  if (g.n <= 64) {
    hipLaunchKernelGGL((gemm_split_kernel<TA, TB, 128, 64, NS, VEC>), dim3(cdiv(g.n, 64), cdiv(g.m, 128), g.batch * sp), dim3(256), 0, s, g);
  } else if ((int64_t)cdiv(g.n, 128) * cdiv(g.m, 128) * g.batch * sp < 256 && g.m > 64) {
    // too few 128 x 128 tiles for 256 CUs (the 4096-row linear layers): 64-row tiles
    hipLaunchKernelGGL((gemm_split_kernel<TA, TB, 64, 128, NS, VEC>), dim3(cdiv(g.n, 128), cdiv(g.m, 64), g.batch * sp), dim3(256), 0, s, g);
  } else {
    hipLaunchKernelGGL((gemm_split_kernel<TA, TB, 128, 128, NS, VEC>), dim3(cdiv(g.n, 128), cdiv(g.m, 128), g.batch * sp), dim3(256), 0, s, g);
  }
  if (sp > 1) hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv((int64_t)g.m * g.n * g.batch, 256)), dim3(256), 0, s, g);
}

template <bool TA, bool TB>
static void gemm_launch(const gims_gemm& g, hipStream_t s) {
  const bool vec = (g.flags & 3) == 3;
  if (g.precision == GIMS_PREC_BF16X6) {
    if (vec) gemm_launch_ns<TA, TB, 3, true>(g, s);
    else gemm_launch_ns<TA, TB, 3, false>(g, s);
  } else {
    if (vec) gemm_launch_ns<TA, TB, 2, true>(g, s);
    else gemm_launch_ns<TA, TB, 2, false>(g, s);
  }
}

// ------------------------------------------------------------------------------------------------ BatchNorm1d, train mode
// Rows are grouped in up to 8 SEGMENTS (one per call of the module in the reference: image 0 of the batch, image 1 of the
// batch); statistics are per (segment, channel).  A block covers 256 rows x 32 channels.
constexpr int BN_RB = 256;

__device__ __forceinline__ bool seg_of_block(const gims_segments& sg, int blk, int& seg, int& r0, int& nr) {
  for (int s = 0; s < sg.n; ++s) {
    const int nb = (sg.rows[s] + BN_RB - 1) / BN_RB;
    if (blk < nb) {
      seg = s;
      r0 = sg.off[s] + blk * BN_RB;
      nr = min(BN_RB, sg.rows[s] - blk * BN_RB);
      return true;
    }
    blk -= nb;
  }
  return false;
}
static int seg_blocks(const gims_segments& sg) {
  int nb = 0;
  for (int s = 0; s < sg.n; ++s) nb += (sg.rows[s] + BN_RB - 1) / BN_RB;
  return nb;
}
__device__ __forceinline__ int seg_first_block(const gims_segments& sg, int seg) {
  int nb = 0;
  for (int s = 0; s < seg; ++s) nb += (sg.rows[s] + BN_RB - 1) / BN_RB;
  return nb;
}

// partial[blk][c] = (mean, M2) of the block's rows
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, int64_t ld, int c, gims_segments sg, float* __restrict__ partial) {
  __shared__ float red[8][32];
  int seg, r0, nr;
  if (!seg_of_block(sg, blockIdx.y, seg, r0, nr)) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, ch = blockIdx.x * 32 + tx;
  const bool on = ch < c;
  float s = 0.f;
  if (on)
    for (int r = ty; r < nr; r += 8) s += x[(int64_t)(r0 + r) * ld + ch];
  red[ty][tx] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) tot += red[i][tx];
  const float mean = tot / (float)nr;
  __syncthreads();
  float q = 0.f;
  if (on)
    for (int r = ty; r < nr; r += 8) {
      const float d = x[(int64_t)(r0 + r) * ld + ch] - mean;
      q = fmaf(d, d, q);
    }
  red[ty][tx] = q;
  __syncthreads();
  if (ty == 0 && on) {
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) m2 += red[i][tx];
    partial[((int64_t)blockIdx.y * c + ch) * 2] = mean;
    partial[((int64_t)blockIdx.y * c + ch) * 2 + 1] = m2;
  }
}

// Chan's combination of the block partials of one segment, in block order
__device__ __forceinline__ void bn_combine(const float* __restrict__ partial, int c, int ch, const gims_segments& sg, int seg, float& mean, float& m2) {
  const int b0 = seg_first_block(sg, seg), nb = (sg.rows[seg] + BN_RB - 1) / BN_RB;
  float n = 0.f;
  mean = 0.f;
  m2 = 0.f;
  for (int b = 0; b < nb; ++b) {
    const float nb_rows = (float)min(BN_RB, sg.rows[seg] - b * BN_RB);
    const float mb = partial[((int64_t)(b0 + b) * c + ch) * 2], qb = partial[((int64_t)(b0 + b) * c + ch) * 2 + 1];
    const float d = mb - mean, nn = n + nb_rows;
    mean += d * (nb_rows / nn);
    m2 += qb + d * d * (n * nb_rows / nn);
    n = nn;
  }
}

// y = [relu](gamma * (x - mean) / sqrt(var + eps) + beta); saves mean / invstd per (segment, channel); block 0 updates the running
// statistics segment by segment (momentum, unbiased variance: torch.nn.BatchNorm1d)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int64_t ld, int c, gims_segments sg, const float* __restrict__ partial,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                                       float* __restrict__ run_mean, float* __restrict__ run_var, float* __restrict__ save,
                                                       float* __restrict__ y, int64_t ldy, int relu) {
  __shared__ float sm[32], si[32];
  int seg, r0, nr;
  if (!seg_of_block(sg, blockIdx.y, seg, r0, nr)) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, ch = blockIdx.x * 32 + tx;
  const bool on = ch < c;
  if (ty == 0 && on) {
    float mean, m2;
    bn_combine(partial, c, ch, sg, seg, mean, m2);
    const float var = m2 / (float)sg.rows[seg];
    sm[tx] = mean;
    si[tx] = 1.f / sqrtf(var + eps);
  }
  if (blockIdx.y == 0 && ty == 1 && on) {
    float rm = run_mean ? run_mean[ch] : 0.f, rv = run_var ? run_var[ch] : 0.f;
    for (int s = 0; s < sg.n; ++s) {
      float mean, m2;
      bn_combine(partial, c, ch, sg, s, mean, m2);
      const float n = (float)sg.rows[s];
      save[((int64_t)s * c + ch) * 2] = mean;
      save[((int64_t)s * c + ch) * 2 + 1] = 1.f / sqrtf(m2 / n + eps);
      rm = (1.f - momentum) * rm + momentum * mean;
      rv = (1.f - momentum) * rv + momentum * (m2 / fmaxf(n - 1.f, 1.f));
    }
    if (run_mean) run_mean[ch] = rm;
    if (run_var) run_var[ch] = rv;
  }
  __syncthreads();
  if (!on) return;
  const float mean = sm[tx], inv = si[tx], gm = gamma[ch], bt = beta[ch];
  for (int r = ty; r < nr; r += 8) {
    const float xh = (x[(int64_t)(r0 + r) * ld + ch] - mean) * inv;
    float v = fmaf(xh, gm, bt);
    if (relu) v = fmaxf(v, 0.f);
    y[(int64_t)(r0 + r) * ldy + ch] = v;
  }
}

// backward: partial[blk][c] = (sum dyr, sum dyr * xhat), dyr = dy masked by the ReLU that followed the norm
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ x, int64_t ld, const float* __restrict__ dy, int64_t ldd, int c,
                                                             gims_segments sg, const float* __restrict__ save, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int relu, float* __restrict__ partial) {
  __shared__ float red[2][8][32];
  int seg, r0, nr;
  if (!seg_of_block(sg, blockIdx.y, seg, r0, nr)) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, ch = blockIdx.x * 32 + tx;
  const bool on = ch < c;
  float s1 = 0.f, s2 = 0.f;
  if (on) {
    const float mean = save[((int64_t)seg * c + ch) * 2], inv = save[((int64_t)seg * c + ch) * 2 + 1], gm = gamma[ch], bt = beta[ch];
    for (int r = ty; r < nr; r += 8) {
      const float xh = (x[(int64_t)(r0 + r) * ld + ch] - mean) * inv;
      float d = dy[(int64_t)(r0 + r) * ldd + ch];
      if (relu && !(fmaf(xh, gm, bt) > 0.f)) d = 0.f;
      s1 += d;
      s2 = fmaf(d, xh, s2);
    }
  }
  red[0][ty][tx] = s1;
  red[1][ty][tx] = s2;
  __syncthreads();
  if (ty == 0 && on) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a += red[0][i][tx];
      b += red[1][i][tx];
    }
    partial[((int64_t)blockIdx.y * c + ch) * 2] = a;
    partial[((int64_t)blockIdx.y * c + ch) * 2 + 1] = b;
  }
}

__device__ __forceinline__ void bn_sum2(const float* __restrict__ partial, int c, int ch, const gims_segments& sg, int seg, float& s1, float& s2) {
  const int b0 = seg_first_block(sg, seg), nb = (sg.rows[seg] + BN_RB - 1) / BN_RB;
  s1 = 0.f;
  s2 = 0.f;
  for (int b = 0; b < nb; ++b) {
    s1 += partial[((int64_t)(b0 + b) * c + ch) * 2];
    s2 += partial[((int64_t)(b0 + b) * c + ch) * 2 + 1];
  }
}

// dx = gamma * invstd * (dyr - mean(dyr) - xhat * mean(dyr * xhat)); block 0: dgamma = sum dyr * xhat, dbeta = sum dyr over all segments
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, int64_t ld, const float* __restrict__ dy, int64_t ldd, int c,
                                                           gims_segments sg, const float* __restrict__ save, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int relu, const float* __restrict__ partial,
                                                           float* __restrict__ dx, int64_t ldx, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float a1[32], a2[32];
  int seg, r0, nr;
  if (!seg_of_block(sg, blockIdx.y, seg, r0, nr)) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, ch = blockIdx.x * 32 + tx;
  const bool on = ch < c;
  if (ty == 0 && on) {
    float s1, s2;
    bn_sum2(partial, c, ch, sg, seg, s1, s2);
    a1[tx] = s1 / (float)sg.rows[seg];
    a2[tx] = s2 / (float)sg.rows[seg];
  }
  if (blockIdx.y == 0 && ty == 1 && on) {
    float g1 = 0.f, g2 = 0.f;
    for (int s = 0; s < sg.n; ++s) {
      float s1, s2;
      bn_sum2(partial, c, ch, sg, s, s1, s2);
      g1 += s1;
      g2 += s2;
    }
    dbeta[ch] = g1;
    dgamma[ch] = g2;
  }
  __syncthreads();
  if (!on) return;
  const float mean = save[((int64_t)seg * c + ch) * 2], inv = save[((int64_t)seg * c + ch) * 2 + 1], gm = gamma[ch], bt = beta[ch];
  const float m1 = a1[tx], m2 = a2[tx];
  for (int r = ty; r < nr; r += 8) {
    const float xh = (x[(int64_t)(r0 + r) * ld + ch] - mean) * inv;
    float d = dy[(int64_t)(r0 + r) * ldd + ch];
    if (relu && !(fmaf(xh, gm, bt) > 0.f)) d = 0.f;
    dx[(int64_t)(r0 + r) * ldx + ch] = gm * inv * (d - m1 - xh * m2);
  }
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
// The reference's LayerNorm (gmatcher.py:74-85: over the channels of a point, UNBIASED std, eps added to the std) followed by ReLU
// (gmatcher.py:19-23), reverse pass.  With d = x - mean, sd = sqrt(sum d^2 / (c - 1)), s = sd + eps, xhat = d / s,
// y = a2 xhat + b2 and g = dy masked by y > 0:   dx_i = a2_i g_i / s - mean_j(a2_j g_j) / s - d_i / ((c - 1) sd s^2) sum_j a2_j g_j d_j.
// One wave per row (c <= 512).  Also writes g and g * xhat ([rows][c] each): their column sums are d b2 and d a2.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy, int64_t ldd,
                                                            int64_t rows, int c, const float* __restrict__ a2, const float* __restrict__ b2, float eps,
                                                            int relu, float* __restrict__ dx, int64_t ldo, float* __restrict__ gb, float* __restrict__ ga) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  float v[8], g[8];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int ch = lane + 64 * k;
    v[k] = ch < c ? xr[ch] : 0.f;
    s += v[k];
  }
  const float mean = wave_sum(s) / (float)c;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    v[k] = lane + 64 * k < c ? v[k] - mean : 0.f;          // v <- d
    q += v[k] * v[k];
  }
  const float sd = sqrtf(wave_sum(q) / (float)(c - 1));
  const float sv = sd + eps, inv = 1.f / sv;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int ch = lane + 64 * k;
    g[k] = 0.f;
    if (ch < c) {
      const float xh = v[k] * inv, a = a2[ch];
      float d = dy[row * ldd + ch];
      if (relu && !(fmaf(a, xh, b2[ch]) > 0.f)) d = 0.f;
      gb[row * c + ch] = d;
      ga[row * c + ch] = d * xh;
      g[k] = a * d;
      s1 += g[k];
      s2 = fmaf(g[k], v[k], s2);
    }
  }
  s1 = wave_sum(s1) / (float)c;
  s2 = wave_sum(s2);
  const float k2 = sd > 0.f ? s2 / ((float)(c - 1) * sd * sv * sv) : 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int ch = lane + 64 * k;
    if (ch < c) dx[row * ldo + ch] = (g[k] - s1) * inv - v[k] * k2;
  }
}

// ------------------------------------------------------------------------------------------------ softmax rows
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sh) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
  return r;
}

// in place: s[row][0..cols) <- softmax (gmatcher.py:37); one block per row, blockIdx.y = batch
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, int64_t ld, int64_t stride, int cols) {
  __shared__ float sh[4];
  float* p = s + (int64_t)blockIdx.y * stride + (int64_t)blockIdx.x * ld;
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < cols; j += 256) mx = fmaxf(mx, p[j]);
  mx = block_reduce(mx, true, sh);
  float sum = 0.f;
  for (int j = threadIdx.x; j < cols; j += 256) {
    const float e = __expf(p[j] - mx);
    p[j] = e;
    sum += e;
  }
  sum = block_reduce(sum, false, sh);
  const float inv = 1.f / sum;
  for (int j = threadIdx.x; j < cols; j += 256) p[j] *= inv;
}

// in place on dp: dS = P * (dP - sum_j dP_j P_j)
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ prob, float* __restrict__ dp, int64_t ld, int64_t stride, int cols) {
  __shared__ float sh[4];
  const float* p = prob + (int64_t)blockIdx.y * stride + (int64_t)blockIdx.x * ld;
  float* d = dp + (int64_t)blockIdx.y * stride + (int64_t)blockIdx.x * ld;
  float dot = 0.f;
  for (int j = threadIdx.x; j < cols; j += 256) dot = fmaf(p[j], d[j], dot);
  dot = block_reduce(dot, false, sh);
  for (int j = threadIdx.x; j < cols; j += 256) d[j] = p[j] * (d[j] - dot);
}

// ------------------------------------------------------------------------------------------------ column sums
// out[ch] = beta * out[ch] + sum_rows x[row][ch].  A workgroup covers 128 rows x 64 channels (16 row lanes x 16 float4 column
// groups), writes its partial sums, and the last workgroup of a channel group to finish adds the partials in block order
// (deterministic); counters: zero on entry, zero on exit.  c and the pitch must be multiples of 4 (vector loads).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ld, int64_t rows, int c, float beta, float* __restrict__ out,
                                                     float* __restrict__ work, unsigned* __restrict__ counters) {
  __shared__ f32x4 red[16][16];
  __shared__ bool last;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, ch = blockIdx.x * 64 + tx * 4;
  const bool on = ch < c;
  const int64_t r0 = (int64_t)blockIdx.y * 128, r1 = min(rows, r0 + 128);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (on)
    for (int64_t r = r0 + ty; r < r1; r += 16) s += *(const f32x4*)(x + r * ld + ch);
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && on) {
    f32x4 tot = red[0][tx];
#pragma unroll
    for (int i = 1; i < 16; ++i) tot += red[i][tx];
    float* w = work + (int64_t)blockIdx.y * c + ch;
#pragma unroll
    for (int q = 0; q < 4; ++q) __hip_atomic_store(w + q, tot[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(counters + blockIdx.x, 1u) == gridDim.y - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  {
    // 64 channels x 4 lanes over the row blocks; each lane keeps 8 loads in flight (a serial chain of device-scope loads costs
    // ~0.7 us per partial); lanes are folded in order, so the sum is still a fixed sequence
    float* fr = (float*)red;
    const int cl = threadIdx.x & 63, lane4 = threadIdx.x >> 6, cc = blockIdx.x * 64 + cl;
    float tot = 0.f;
    if (cc < c)
      for (unsigned b0 = lane4 * 8; b0 < gridDim.y; b0 += 32) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
          v[q] = b0 + q < gridDim.y ? __hip_atomic_load(work + (int64_t)(b0 + q) * c + cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) tot += v[q];
      }
    __syncthreads();
    fr[lane4 * 64 + cl] = tot;
    __syncthreads();
    if (lane4 == 0 && cc < c) {
      const float t = (fr[cl] + fr[64 + cl]) + (fr[128 + cl] + fr[192 + cl]);
      out[cc] = beta != 0.f ? fmaf(beta, out[cc], t) : t;
    }
  }
  if (threadIdx.x == 0) counters[blockIdx.x] = 0;
}

// ------------------------------------------------------------------------------------------------ small element-wise kernels
__global__ __launch_bounds__(256) void ew_kernel(int op, float* __restrict__ out, int64_t ldo, const float* __restrict__ a, int64_t lda,
                                                 const float* __restrict__ b, int64_t ldb, int64_t rows, int cols, float alpha) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols) return;
  const int64_t r = i / cols;
  const int cc = (int)(i - r * cols);
  const float av = a[r * lda + cc];
  float v;
  switch (op) {
    case GIMS_EW_ADD: v = av + alpha * b[r * ldb + cc]; break;
    case GIMS_EW_RELU_MASK: v = b[r * ldb + cc] > 0.f ? av : 0.f; break;
    case GIMS_EW_RELU: v = fmaxf(av, 0.f); break;
    case GIMS_EW_ACC: v = out[r * ldo + cc] + alpha * av; break;
    default: v = alpha * av; break;
  }
  out[r * ldo + cc] = v;
}

// dst[i0*d0 + i1*d1 + i2*d2] (=|+=) src[i0*s0 + i1*s1 + i2*s2]
__global__ __launch_bounds__(256) void permute3_kernel(float* __restrict__ dst, const float* __restrict__ src, int n0, int n1, int n2, int64_t d0, int64_t d1,
                                                       int64_t d2, int64_t s0, int64_t s1, int64_t s2, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)n0 * n1 * n2) return;
  const int i2 = (int)(i % n2), i1 = (int)((i / n2) % n1), i0 = (int)(i / ((int64_t)n1 * n2));
  const float v = src[i0 * s0 + i1 * s1 + i2 * s2];
  float* o = dst + i0 * d0 + i1 * d1 + i2 * d2;
  *o = accumulate ? *o + v : v;
}

// All head-interleave permutations of ONE attention layer in one launch (they were 7 + 7 launches of ~5 us per layer):
// forward = false: proj weights / biases (reference order: channel = d * H + h) -> wqkv [3D][D], bqkv [3D] (head-contiguous rows),
//                  merge weight -> wm [D][D] (head-contiguous columns);
// forward = true : the same maps applied to gradients, from the packed layout back to the parameters' layout.
struct HeadPack {
  float* pw[3]; float* pb[3]; float* mw;        // parameter-layout tensors: proj weights [D][D], proj biases [D], merge weight [D][D]
  float* wqkv; float* bqkv; float* wm;           // packed tensors
  int d, heads;
};
__global__ __launch_bounds__(256) void head_pack_kernel(HeadPack hp, int to_params) {
  const int D = hp.d, H = hp.heads, dh = D / H;
  const int64_t nw = (int64_t)3 * D * D, nb = 3 * D, nm = (int64_t)D * D;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < nw) {                                  // packed row r = j * D + h * dh + dd  <->  parameter row dd * H + h of projection j
    const int col = (int)(idx % D), r = (int)(idx / D), j = r / D, rr = r - j * D, h = rr / dh, dd = rr - h * dh;
    float* par = hp.pw[j] + (int64_t)(dd * H + h) * D + col;
    float* pk = hp.wqkv + idx;
    if (to_params) *par = *pk; else *pk = *par;
  } else if (idx < nw + nb) {
    const int r = (int)(idx - nw), j = r / D, rr = r - j * D, h = rr / dh, dd = rr - h * dh;
    float* par = hp.pb[j] + dd * H + h;
    float* pk = hp.bqkv + r;
    if (to_params) *par = *pk; else *pk = *par;
  } else if (idx < nw + nb + nm) {                 // packed column h * dh + dd  <->  parameter column dd * H + h
    const int64_t e = idx - nw - nb;
    const int n = (int)(e / D), c = (int)(e - (int64_t)n * D), h = c / dh, dd = c - h * dh;
    float* par = hp.mw + (int64_t)n * D + dd * H + h;
    float* pk = hp.wm + e;
    if (to_params) *par = *pk; else *pk = *par;
  }
}

// transposed mean aggregation: out_j = sum over in-neighbours i of j (symmetric CSR) of g_i / deg_i  -- the gradient of
// mean_{j in N(i)} h_j with respect to h (SAGEConv 'mean', gmatcher.py:149-158)
__global__ __launch_bounds__(256) void sage_mean_t_kernel(const float* __restrict__ g, int64_t ldg, const int32_t* __restrict__ indptr,
                                                          const int32_t* __restrict__ indices, int n, int c, float* __restrict__ out, int64_t ldo) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int node = blockIdx.x * 4 + wave;
  if (node >= n) return;
  const int beg = indptr[node], end = indptr[node + 1];
  for (int q = lane; 4 * q < c; q += 64) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = beg; e < end; ++e) {
      const int i = indices[e];
      const float d = (float)(indptr[i + 1] - indptr[i]);
      const float4 x = *(const float4*)(g + (int64_t)i * ldg + 4 * q);
      s.x += x.x / d; s.y += x.y / d; s.z += x.z / d; s.w += x.w / d;
    }
    *(float4*)(out + (int64_t)node * ldo + 4 * q) = s;
  }
}

// normalize_keypoints (gmatcher.py:26-33) as its own tensor: (k - centre) / scale per image
__global__ __launch_bounds__(256) void normalize_kpts_kernel(const float* __restrict__ kpts, const float* __restrict__ norm3, const int32_t* __restrict__ seg,
                                                             int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * n) return;
  const float* nm = norm3 + 3 * seg[i >> 1];
  out[i] = (kpts[i] - nm[i & 1]) / nm[2];
}

}  // namespace gims

// ================================================================================================ C ABI
using namespace gims;

extern "C" int gims_gemm_f32(const gims_gemm* gp, void* stream) {
  GIMS_CHECK_ARG(gp && gp->a && gp->b && gp->c && gp->m >= 0 && gp->n >= 0 && gp->k >= 0 && gp->batch >= 1 && gp->batch <= 65535,
                 "gims_gemm_f32: bad arguments");
  GIMS_CHECK_ARG(gp->precision == GIMS_PREC_BF16X3 || gp->precision == GIMS_PREC_BF16X6, "gims_gemm_f32: precision is GIMS_PREC_BF16X3 or GIMS_PREC_BF16X6");
  gims_gemm g = *gp;
  if (g.m == 0 || g.n == 0) return GIMS_OK;
  GIMS_CHECK_ARG(g.lda >= (g.ta ? g.m : g.k) && g.ldb >= (g.tb ? g.n : g.k) && g.ldc >= g.n && (!g.residual || g.ldr >= g.n),
                 "gims_gemm_f32: leading dimension smaller than the row it strides");
  auto vec_ok = [&](const float* p, int64_t ld, int64_t st) { return ((uintptr_t)p & 15) == 0 && (ld & 3) == 0 && (g.batch == 1 || (st & 3) == 0); };
  g.flags = (vec_ok(g.a, g.lda, g.sa) ? 1 : 0) | (vec_ok(g.b, g.ldb, g.sb) ? 2 : 0);
  if (((uintptr_t)g.c & 15) == 0 && (g.ldc & 3) == 0 && (g.batch == 1 || (g.sc & 3) == 0) &&
      (!g.residual || (((uintptr_t)g.residual & 15) == 0 && (g.ldr & 3) == 0 && (g.batch == 1 || (g.sr & 3) == 0))))
    g.flags |= 8;
  if (g.work && ((uintptr_t)g.work & 15) == 0 && (g.n & 3) == 0) g.flags |= 16;      // (m n is then a multiple of 4 as well: every partial tile starts aligned)
  static const bool prof_on = getenv("GIMS_GEMM_PROF") != nullptr;
  if (prof_on) g.flags |= 4;
  // split-K when the output has too few tiles to fill the chip and k is long (the weight gradients: k = keypoint rows):
  // every split covers >= 128 k, partial sums go through the caller's workspace and are added in split order
  g.splits = 1;
  if (g.work && g.k >= 512) {
    const int64_t tiles = (int64_t)cdiv(g.n, g.n <= 64 ? 64 : 128) * cdiv(g.m, 128) * g.batch;
    if (tiles < 128) {
      // count 64-row tiles where the launch can use them (n > 64): half the splits fill the chip just as well, and the partial
      // sums the fold has to read halve with them
      const int64_t fine = g.n > 64 && g.m > 64 ? (int64_t)cdiv(g.n, 128) * cdiv(g.m, 64) * g.batch : tiles;
      int sp = (int)((384 + fine - 1) / fine);
      sp = sp < g.k / 128 ? sp : g.k / 128;
      const int64_t cap = g.work_floats / ((int64_t)g.m * g.n * g.batch);
      sp = sp < cap ? sp : (int)cap;
      if (sp > 1) g.splits = sp;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  if (!g.ta && !g.tb) gemm_launch<false, false>(g, s);
  else if (!g.ta && g.tb) gemm_launch<false, true>(g, s);
  else if (g.ta && !g.tb) gemm_launch<true, false>(g, s);
  else gemm_launch<true, true>(g, s);
  GIMS_LAUNCH_CHECK();
  if (prof_on) {
    unsigned long long h[8];
    GIMS_HIP(hipStreamSynchronize(s));
    GIMS_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm_prof), sizeof(h)));
    fprintf(stderr, "gemm m %d n %d k %d b %d ta %d tb %d splits %d: prologue %llu  main loop %llu (%d k-tiles: issue+barrier %llu, lds-read+mfma %llu, barrier %llu, convert+lds-write %llu)  epilogue %llu  (cycles of one workgroup)\n",
            g.m, g.n, g.k, g.batch, g.ta, g.tb, g.splits, h[1] - h[0], h[2] - h[1], (g.k / (g.splits > 1 ? g.splits : 1) + 31) / 32, h[4], h[5], h[6], h[7], h[3] - h[2]);
  }
  return GIMS_OK;
}

static int check_segments(const gims_segments* sg, const char* who) {
  GIMS_CHECK_ARG(sg && sg->n >= 1 && sg->n <= 8, "%s: 1..8 row segments", who);
  for (int s = 0; s < sg->n; ++s) GIMS_CHECK_ARG(sg->rows[s] >= 1 && sg->off[s] >= 0, "%s: empty or negative row segment", who);
  return GIMS_OK;
}

extern "C" size_t gims_batchnorm_workspace_floats(const gims_segments* sg, int32_t c) {
  if (!sg || sg->n < 1 || sg->n > 8 || c <= 0) return 0;
  return (size_t)seg_blocks(*sg) * (size_t)c * 2;
}

extern "C" int gims_batchnorm_train_forward(const float* x, int64_t ld, int32_t c, const gims_segments* sg, const float* gamma, const float* beta, float eps,
                                            float momentum, float* running_mean, float* running_var, float* save, float* y, int64_t ldy, int32_t relu,
                                            float* work, void* stream) {
  GIMS_CHECK_ARG(x && gamma && beta && save && y && work && c > 0 && ld >= c && ldy >= c, "gims_batchnorm_train_forward: bad arguments");
  if (int rc = check_segments(sg, "gims_batchnorm_train_forward")) return rc;
  const dim3 grid(cdiv(c, 32), seg_blocks(*sg));
  hipLaunchKernelGGL(bn_partial_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ld, c, *sg, work);
  hipLaunchKernelGGL(bn_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ld, c, *sg, (const float*)work, gamma, beta, eps, momentum,
                     running_mean, running_var, save, y, ldy, relu);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_batchnorm_train_backward(const float* x, int64_t ld, const float* dy, int64_t ldd, int32_t c, const gims_segments* sg, const float* save,
                                             const float* gamma, const float* beta, int32_t relu, float* dx, int64_t ldx, float* dgamma, float* dbeta,
                                             float* work, void* stream) {
  GIMS_CHECK_ARG(x && dy && save && gamma && beta && dx && dgamma && dbeta && work && c > 0 && ld >= c && ldd >= c && ldx >= c,
                 "gims_batchnorm_train_backward: bad arguments");
  if (int rc = check_segments(sg, "gims_batchnorm_train_backward")) return rc;
  const dim3 grid(cdiv(c, 32), seg_blocks(*sg));
  hipLaunchKernelGGL(bn_bwd_partial_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ld, dy, ldd, c, *sg, save, gamma, beta, relu, work);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ld, dy, ldd, c, *sg, save, gamma, beta, relu, (const float*)work, dx,
                     ldx, dgamma, dbeta);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_softmax_rows(float* s, int64_t ld, int64_t rows, int32_t cols, int32_t batch, int64_t stride, void* stream) {
  GIMS_CHECK_ARG(s && ld >= cols && rows >= 0 && cols >= 1 && batch >= 1 && batch <= 65535, "gims_softmax_rows: bad arguments");
  if (rows == 0) return GIMS_OK;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows, batch), dim3(256), 0, (hipStream_t)stream, s, ld, stride, cols);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_softmax_rows_backward(const float* prob, float* dp, int64_t ld, int64_t rows, int32_t cols, int32_t batch, int64_t stride, void* stream) {
  GIMS_CHECK_ARG(prob && dp && ld >= cols && rows >= 0 && cols >= 1 && batch >= 1 && batch <= 65535, "gims_softmax_rows_backward: bad arguments");
  if (rows == 0) return GIMS_OK;
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)rows, batch), dim3(256), 0, (hipStream_t)stream, prob, dp, ld, stride, cols);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" size_t gims_colsum_workspace_floats(int64_t rows, int32_t c) {
  if (rows <= 0 || c <= 0) return 0;
  return 64 + (size_t)cdiv(rows, 128) * (size_t)c;          // [64 counters (fixed place, zero between calls)][row blocks][c] partials
}

extern "C" int gims_colsum(const float* x, int64_t ld, int64_t rows, int32_t c, float beta, float* out, float* work, void* stream) {
  GIMS_CHECK_ARG(x && out && work && rows >= 1 && c >= 4 && c <= 4096 && ld >= c && (c & 3) == 0 && (ld & 3) == 0 && ((uintptr_t)x & 15) == 0,
                 "gims_colsum: bad arguments (c <= 4096; c, pitch multiples of 4, 16-byte aligned)");
  const int nrb = cdiv(rows, 128);
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(c, 64), nrb), dim3(256), 0, (hipStream_t)stream, x, ld, rows, c, beta, out, work + 64, (unsigned*)work);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_elementwise(int32_t op, float* out, int64_t ldo, const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int32_t cols,
                                float alpha, void* stream) {
  GIMS_CHECK_ARG(out && a && rows >= 0 && cols >= 1 && ldo >= cols && lda >= cols, "gims_elementwise: bad arguments");
  GIMS_CHECK_ARG(op >= GIMS_EW_SCALE && op <= GIMS_EW_ACC, "gims_elementwise: unknown operation %d", op);
  GIMS_CHECK_ARG((op != GIMS_EW_ADD && op != GIMS_EW_RELU_MASK) || (b && ldb >= cols), "gims_elementwise: this operation needs a second operand");
  if (rows == 0) return GIMS_OK;
  hipLaunchKernelGGL(ew_kernel, dim3((unsigned)cdiv(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, op, out, ldo, a, lda, b, ldb, rows, cols, alpha);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_permute3(float* dst, const float* src, int32_t n0, int32_t n1, int32_t n2, int64_t d0, int64_t d1, int64_t d2, int64_t s0, int64_t s1,
                             int64_t s2, int32_t accumulate, void* stream) {
  GIMS_CHECK_ARG(dst && src && n0 >= 1 && n1 >= 1 && n2 >= 1, "gims_permute3: bad arguments");
  hipLaunchKernelGGL(permute3_kernel, dim3((unsigned)cdiv((int64_t)n0 * n1 * n2, 256)), dim3(256), 0, (hipStream_t)stream, dst, src, n0, n1, n2, d0, d1, d2,
                     s0, s1, s2, accumulate);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_sage_mean_transposed(const float* g, int64_t ldg, const int32_t* indptr, const int32_t* indices, int32_t n, int32_t c, float* out,
                                         int64_t ldo, void* stream) {
  GIMS_CHECK_ARG(g && indptr && indices && out && n >= 0 && c >= 4 && (c % 4) == 0 && (ldg % 4) == 0 && (ldo % 4) == 0,
                 "gims_sage_mean_transposed: bad arguments (c, pitches multiples of 4)");
  if (n == 0) return GIMS_OK;
  hipLaunchKernelGGL(sage_mean_t_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, g, ldg, indptr, indices, n, c, out, ldo);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_normalize_keypoints(const float* kpts, const float* norm3, const int32_t* seg_of_row, int64_t n, float* out, void* stream) {
  GIMS_CHECK_ARG(kpts && norm3 && seg_of_row && out && n >= 0, "gims_normalize_keypoints: bad arguments");
  if (n == 0) return GIMS_OK;
  hipLaunchKernelGGL(normalize_kpts_kernel, dim3((unsigned)cdiv(2 * n, 256)), dim3(256), 0, (hipStream_t)stream, kpts, norm3, seg_of_row, n, out);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_layernorm_backward(const float* x, int64_t ldx, const float* dy, int64_t ldd, int64_t rows, int32_t c, const float* a2, const float* b2,
                                       float eps, int32_t relu, float* dx, int64_t ldo, float* g_bias, float* g_scale, void* stream) {
  GIMS_CHECK_ARG(x && dy && a2 && b2 && dx && g_bias && g_scale && rows >= 0 && c >= 2 && c <= 512 && ldx >= c && ldd >= c && ldo >= c,
                 "gims_layernorm_backward: bad arguments (2 <= c <= 512)");
  if (rows == 0) return GIMS_OK;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, dy, ldd, rows, c, a2, b2, eps, relu, dx,
                     ldo, g_bias, g_scale);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_head_pack(float* const* proj_w, float* const* proj_b, float* merge_w, float* wqkv, float* bqkv, float* wm, int32_t d, int32_t heads,
                              int32_t to_params, void* stream) {
  GIMS_CHECK_ARG(proj_w && proj_b && merge_w && wqkv && bqkv && wm && d > 0 && heads > 0 && (d % heads) == 0, "gims_head_pack: bad arguments");
  HeadPack hp;
  for (int j = 0; j < 3; ++j) {
    GIMS_CHECK_ARG(proj_w[j] && proj_b[j], "gims_head_pack: projection %d has a null pointer", j);
    hp.pw[j] = proj_w[j];
    hp.pb[j] = proj_b[j];
  }
  hp.mw = merge_w; hp.wqkv = wqkv; hp.bqkv = bqkv; hp.wm = wm; hp.d = d; hp.heads = heads;
  const int64_t total = (int64_t)3 * d * d + 3 * d + (int64_t)d * d;
  hipLaunchKernelGGL(head_pack_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, hp, to_params);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

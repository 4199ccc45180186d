// Exact-class dense contraction on bf16 MFMA: C = scale * A W^T with both f32 operands split THREE ways,
//     a = a1 + a2 + a3   (bf16 each; 8 + 8 + 8 significant bits: the split of an f32 value is exact)
//     a . w ~ a1 w1 + (a1 w2 + a2 w1) + (a1 w3 + a2 w2 + a3 w1)            six v_mfma_f32_32x32x16_bf16 per product,
// the dropped terms are <= 2^-24 relative -- the accuracy of an f32 GEMM (measured: 6e-8 against float64 on unit-norm
// descriptors, an f32 BLAS GEMM 6e-7) at 6/16 of the cost of the exact-f32 MFMA (v_mfma_f32_32x32x2_f32 runs at 1/16 of
// the bf16 rate).  Used for the two N x N contractions whose VALUES are compared or thresholded downstream: the cosine
// similarity matrix of the adaptive graph (agc.py:390) and the score matrix (gmatcher.py:274).
//
// Operand layout "SPL3": logical [rows][K] f32 -> bf16 [rows][3K]; per 32-channel block 32 x a1, 32 x a2, 32 x a3
// (192 bytes).  Kernel: PERSISTENT over the tile list of the batch; 256 x 128 output tile, 8 waves (4 x 2, 64 x 64 each), 32-channel stages in a 2-stage LDS ring filled
// by LDS-DMA; a stage row is 12 chunks of 16 bytes, stored ROTATED by (row >> 2) % 12 chunks (the rotation is applied to
// the source chunk index of the DMA and again on the ds_read_b128: conflict-free for the b128 lane groups; 192-byte rows
// without it are 4-way conflicted).  Epilogue: accumulators transposed through LDS so that stores cover whole lines.
#include "common.h"

#include <stdlib.h>

namespace gims {

constexpr int X6_TM = 256, X6_TN = 128, X6_WN = 2, X6_BK = 32;          // 8 waves as 4 x X6_WN, 64 x 64 each
constexpr int X6_ROW = 96;                                   // bf16 elements per stage row (3 planes x 32 channels)
constexpr int X6_STAGE = (X6_TM + X6_TN) * X6_ROW;           // elements per stage (73 728 bytes)
constexpr int X6_PIECES = (X6_TM + X6_TN) * 12 / 64 / 8;     // 16-byte x 64-lane DMA instructions per wave per stage (9)
constexpr int X6_EP_PITCH = X6_TN / X6_WN + 4;
constexpr int X6_LDS_BYTES = 2 * X6_STAGE * 2;               // 147 456 bytes (the epilogue slices, 8 x 8.7 KB, reuse it)
constexpr int X6_MAX_PROBLEMS = 1024, X6_TABLE_BYTES = (X6_MAX_PROBLEMS + 1) * 4;   // tile table behind the ring

template <int N>
__device__ __forceinline__ void x6_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ int x6_rot(int row) { return (row >> 2) % 12; }

// tiles of problem (m, n): all of them, or for a symmetric product (GIMS_LINEAR_UPPER) those not entirely below the
// diagonal -- row by of 256-row tiles keeps the 128-column tiles bx >= 2 by
__device__ __forceinline__ int x6_rows_kept(int ntm, int ntn) { const int r = (ntn + 1) / 2; return ntm < r ? ntm : r; }
__device__ __forceinline__ int x6_upper_before(int by, int ntn) { return by * ntn - by * (by - 1); }   // tiles in rows < by

// PERSISTENT: one workgroup per CU walks the tile list of the whole batch (tile t -> workgroup t mod gridDim.x).  With
// K = 256 a tile is only eight stages; as one-shot workgroups (147 KB of LDS each, so strictly one after the other on a
// CU) the launch, the argument fetch and the un-overlapped first stage cost as much as the K loop, and the workgroups
// of a symmetric product that lie below the diagonal still queued for a CU each just to exit.
__global__ __launch_bounds__(512) void linear_x6_kernel(const gims_linear_args* __restrict__ args, int count) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  int* tstart = (int*)(smem + 2 * X6_STAGE);                    // [count + 1] first tile of every problem
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / X6_WN, wn = wave % X6_WN;
  const int li = lane & 31, lh = lane >> 5;
  for (int i = t; i < count; i += 512) {
    const int ntm = (args[i].m + X6_TM - 1) / X6_TM, ntn = (args[i].n + X6_TN - 1) / X6_TN;
    tstart[i + 1] = (args[i].flags & GIMS_LINEAR_UPPER) ? x6_upper_before(x6_rows_kept(ntm, ntn), ntn) : ntm * ntn;
  }
  __syncthreads();
  if (t == 0) {
    tstart[0] = 0;
    for (int i = 1; i <= count; ++i) tstart[i] += tstart[i - 1];
  }
  __syncthreads();
  const int total = __builtin_amdgcn_readfirstlane(tstart[count]);
  int z = 0;
  for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
  while (tile >= __builtin_amdgcn_readfirstlane(tstart[z + 1])) ++z;
  const gims_linear_args p = args[z];
  int by, bx;
  {
    const int u = tile - __builtin_amdgcn_readfirstlane(tstart[z]);
    const int ntn = (p.n + X6_TN - 1) / X6_TN;
    if (p.flags & GIMS_LINEAR_UPPER) {
      by = 0;
      while (u >= x6_upper_before(by + 1, ntn)) ++by;
      bx = 2 * by + (u - x6_upper_before(by, ntn));
    } else {
      by = u / ntn;
      bx = u - by * ntn;
    }
  }
  const int m0 = by * X6_TM, n0 = bx * X6_TN;
  // diagnostic bits (tools/x6_probe.py only): 0x100 no K loop, 0x200 no epilogue, 0x400 no DMA after the first stage, 0x800 no MFMA
  const int nk = (p.flags & 0x100) ? 0 : p.k / X6_BK;

  // DMA duty of this lane: chunk c = (wave * 9 + i) * 64 + lane of the stage image; row = c / 12, position = c % 12 holds
  // the source chunk (position - rot(row)) mod 12.  Element offsets relative to the k-block start are loop invariant.
  int64_t goff[X6_PIECES];
#pragma unroll
  for (int i = 0; i < X6_PIECES; ++i) {
    const int c = (wave * X6_PIECES + i) * 64 + lane;
    const int row = c / 12, pos = c % 12;
    int g = pos - x6_rot(row);
    g = g < 0 ? g + 12 : g;
    const bool is_a = row < X6_TM;
    int gr = is_a ? m0 + row : n0 + row - X6_TM;
    const int rmax = (is_a ? p.m : p.n) - 1;
    gr = gr < rmax ? gr : rmax;
    goff[i] = (is_a ? (int64_t)gr * p.lda0 : (int64_t)gr * p.ldw) + 8 * g;
  }
  const uint16_t* abase = (const uint16_t*)p.a0;
  const uint16_t* wbase = (const uint16_t*)p.w;
  auto issue = [&](int kt) {
    uint16_t* dst = smem + (kt & 1) * X6_STAGE + wave * X6_PIECES * 512;
#pragma unroll
    for (int i = 0; i < X6_PIECES; ++i) {
      const int c0 = (wave * X6_PIECES + i) * 64;            // wave-uniform: rows of one instruction are all A or all W
      const uint16_t* g = ((c0 / 12) < X6_TM ? abase : wbase) + goff[i] + (int64_t)kt * X6_ROW;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(dst + i * 512), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];   // [n-block][m-block], D^T layout: column = row m (lane & 31), rows = output channels
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: row r of the stage image, chunk (plane * 4 + 2 * s + lh + rot(r)) mod 12
  int arow[2], wrow[2], arot[2], wrot[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    arow[i] = wm * 64 + i * 32 + li;
    wrow[i] = X6_TM + wn * 64 + i * 32 + li;
    arot[i] = x6_rot(arow[i]);
    wrot[i] = x6_rot(wrow[i]);
  }
  issue(0);
  for (int kt = 0; kt < nk; ++kt) {
    x6_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk && !(p.flags & 0x400)) issue(kt + 1);
    if (p.flags & 0x800) continue;
    const uint16_t* st = smem + (kt & 1) * X6_STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[2][3], wf[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          int ca = pl * 4 + 2 * s + lh + arot[i];
          ca = ca >= 12 ? ca - 12 : ca;
          int cw = pl * 4 + 2 * s + lh + wrot[i];
          cw = cw >= 12 ? cw - 12 : cw;
          af[i][pl] = *(const bf16x8*)(st + arow[i] * X6_ROW + 8 * ca);
          wf[i][pl] = *(const bf16x8*)(st + wrow[i] * X6_ROW + 8 * cw);
        }
      // six products per accumulator, smallest terms first; term-major so that consecutive MFMAs hit different accumulators
#pragma unroll
      for (int term = 0; term < 6; ++term) {
        constexpr int WP[6] = {2, 0, 1, 1, 0, 0}, AP[6] = {0, 2, 1, 0, 1, 0};   // (w3 a1) (w1 a3) (w2 a2) (w2 a1) (w1 a2) (w1 a1)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni][WP[term]], af[mi][AP[term]], acc[ni][mi], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: transpose through a wave-private LDS slice (32 rows x 64 columns at a time), row-contiguous stores
  __builtin_amdgcn_s_barrier();
  if (!((p.flags & 0x200) && acc[0][0][0] != 12345.678f)) {
  float* ep = (float*)smem + wave * (32 * X6_EP_PITCH);
  constexpr int LPR = 64 / 8, RPI = 64 / LPR, ITERS = 32 / RPI;     // 8 lanes per row, 8 rows per access, 4 accesses
  const int c8 = (lane % LPR) * 8, rsub = lane / LPR;
  const int col = n0 + wn * 64 + c8;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(float4*)(ep + li * X6_EP_PITCH + ni * 32 + 8 * g + 4 * lh) =
            make_float4(acc[ni][mi][4 * g], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int row = m0 + wm * 64 + mi * 32 + rsub + it * RPI;
      const float* src = ep + (rsub + it * RPI) * X6_EP_PITCH + c8;
      const float4 a0 = *(const float4*)src, a1 = *(const float4*)(src + 4);
      if (row < p.m) {
        float* o = p.out_f32 + (int64_t)row * p.ldc + col;
        if (col + 3 < p.n) *(float4*)o = make_float4(a0.x * p.scale, a0.y * p.scale, a0.z * p.scale, a0.w * p.scale);
        else {
          const float v[4] = {a0.x, a0.y, a0.z, a0.w};
          for (int e = 0; e < 4; ++e)
            if (col + e < p.n) o[e] = v[e] * p.scale;
        }
        if (col + 7 < p.n) *(float4*)(o + 4) = make_float4(a1.x * p.scale, a1.y * p.scale, a1.z * p.scale, a1.w * p.scale);
        else {
          const float v[4] = {a1.x, a1.y, a1.z, a1.w};
          for (int e = 0; e < 4; ++e)
            if (col + 4 + e < p.n) o[4 + e] = v[e] * p.scale;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  }
  __builtin_amdgcn_s_barrier();          // the next tile's first stage overwrites the epilogue slices of other waves
  }
}

// f32 [rows][k] -> SPL3 bf16 [rows][3k]: one thread per (row, channel)
__global__ void split_spl3_kernel(const float* __restrict__ src, int64_t lds, uint16_t* __restrict__ dst, int64_t ldd, int64_t rows, int k) {
  const int64_t total = rows * k;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / k;
    const int c = (int)(idx - r * k);
    const float x = src[r * lds + c];
    const uint16_t h1 = f2bf(x);
    const float r1 = x - bf2f(h1);
    const uint16_t h2 = f2bf(r1);
    const uint16_t h3 = f2bf(r1 - bf2f(h2));
    uint16_t* d = dst + r * ldd + (c >> 5) * 96 + (c & 31);
    d[0] = h1; d[32] = h2; d[64] = h3;
  }
}

int linear_x6_batch_launch(const gims_linear_args* dev_args, int count, int max_m, int max_n, hipStream_t s) {
  GIMS_LDS_ATTR((const void*)linear_x6_kernel, X6_LDS_BYTES + X6_TABLE_BYTES);
  GIMS_CHECK_ARG(count >= 1 && count <= X6_MAX_PROBLEMS, "gims_linear_batch(bf16x6): at most %d problems per launch", X6_MAX_PROBLEMS);
  const int64_t bound = (int64_t)cdiv(max_n, X6_TN) * cdiv(max_m, X6_TM) * count;      // workgroups beyond the tile list exit at once
  hipLaunchKernelGGL(linear_x6_kernel, dim3((int)(bound < 256 ? bound : 256)), dim3(512), X6_LDS_BYTES + X6_TABLE_BYTES, s, dev_args, count);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

int split_spl3_launch(const float* src, int64_t lds, uint16_t* dst, int64_t ldd, int64_t rows, int k, hipStream_t s) {
  int64_t blocks = (rows * k + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(split_spl3_kernel, dim3((int)blocks), dim3(256), 0, s, src, lds, dst, ldd, rows, k);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

}  // namespace gims

extern "C" int gims_split_spl3(const float* src, int64_t lds, uint16_t* dst, int64_t ldd, int64_t rows, int32_t k, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(src && dst && rows >= 0 && k > 0 && (k % 32) == 0 && ldd >= 3 * (int64_t)k && (ldd % 8) == 0 && (((uintptr_t)dst) & 15) == 0,
                 "gims_split_spl3: bad arguments (k %% 32 == 0, ldd >= 3k, 16-byte aligned rows)");
  if (rows == 0) return GIMS_OK;
  return split_spl3_launch(src, lds, dst, ldd, rows, k, (hipStream_t)stream);
}

// On-chip Sinkhorn, 2-D decomposition (round 3): all iterations of log_sinkhorn_iterations (gmatcher.py:41-47) in ONE launch with
// the transport matrix resident in registers + LDS, like ot_resident_kernel (sinkhorn.hip) -- but decomposed so that the
// per-iteration exchanges are small and mostly stay inside one XCD's L2.
//
// ot_resident_kernel gives a workgroup a slab of whole ROWS: row sums are local, but the column sums are an all-reduce of an
// (m+1)-vector over all 128 slabs of a problem -- 16 KB published and 34 KB read per workgroup and iteration, all of it
// through memory-side (cross-XCD) coherence: 62 % of its 13 us iteration.  Here a problem of (n+1) x (m+1) is cut into
//     nx row groups (<= 1024 rows each; one group lives on the CUs of ONE XCD)  x  nc column blocks (<= 128 columns each),
// one workgroup (512 threads, one per CU) per (row group, column block), holding a <= 1024 x 128 block of
// K = exp(Z + u + v) as 16 x 16 tiles per thread (12 tile rows in registers, 4 in LDS).  Per iteration:
//   row sums     partial over the 128 columns -> 4 KB published IN THE XCD'S L2 (plain stores, L1-bypassing loads) ->
//                every workgroup folds 1/nc of the group's rows over the nc blocks, updates u and the cumulative row factor F
//                -> publishes 4 F per fold lane -> every workgroup of the group reads the group's F (4 KB, same L2);
//   column sums  partial over the group's rows for the workgroup's OWN 128 columns -> 512 B published cross-XCD
//                (write-through stores) -> each of the nx workgroups that share the columns reads the other nx - 1 partials
//                and updates v and the cumulative column factor G for its own columns, redundantly and identically.
// So the two wide edges (32 participants) never leave the XCD, and the one edge that crosses XCDs has 4 participants and
// 0.5 KB -- no all-gather of G at all, because a workgroup only ever needs G for its own columns.
// Numerics: the lazy-scaling form of ot_resident_kernel: K of the last derivation is never rewritten, the transport matrix is
// diag(F) K diag(G) with F_i = mu_i / sum_j K_ij G_j and G_j = nu_j / sum_i F_i K_ij taken from the CURRENT iteration's sums alone, and
// u = u(last derivation) + log F, v likewise -- nothing is accumulated from one iteration to the next, so the potentials are one rounding
// away from the factors the iteration balances, and K is re-derived from Z, u, v only WHEN NEEDED.  Fixed summation orders everywhere:
// bitwise deterministic.
// When a derivation is needed (round 5).  K = exp(Z + u + v) is formed in f32: an entry below e^-87 at the derivation is zero from then on,
// whatever the factors do to it later.  That is harmless while the entry stays negligible -- and wrong once F_i G_j has grown by enough to make
// it matter: the dustbin column of a SPARSE pair (u up by 55, v_bin up by 37 over 100 iterations: entries that started at e^-120 end as the
// dominant terms of their rows) cost 1e-2 on the scores without a mid-solve derivation, while dense pairs (growth <= 41) lose nothing.  A fixed
// period of 50 served both at the price of one sweep of Z per solve; now every workgroup flags the iteration at whose end one of ITS factors
// exceeds a bound (log F > 32 or log G > 20: an entry can then have grown by <= 52 + what the lag adds, i.e. anything that can matter, >= e^-26
// of a row total, was >= e^-87 when K was formed), and every fourth iteration and the last one are candidates: a candidate c re-derives iff a
// flag was raised in the iterations (max(c - 8, last derivation), c - 4].  A dense pair derives K once, from the start potentials.  The decision is the same in every workgroup of the problem: a workgroup that has finished
// iteration k has, through the two exchanges of that iteration, seen data that every other workgroup published after finishing iteration k - 1
// (its flag store, fenced, precedes that), so flags of iterations <= c - 3 are final when anybody reads them at the top of iteration c - 1.
// Exchange protocol: no barrier; every exchanged value is >= +0 and carries the parity of its iteration in the sign bit,
// readers re-read until it matches (bounded: a wait that runs out flags status 2 and the caller's rescue re-solves).
// The column edge is double-buffered by iteration parity (its readers are not ordered against its next writer).
// Placement: the host puts the nc workgroups of a row group on blockIdx values that share b % 8 (observed: block b runs on XCD
// b % 8).  Each workgroup CHECKS its XCC id against that; on any mismatch the launch uses write-through stores for the local
// edges too (slower, placement-independent) -- results never depend on where the dispatcher put a workgroup.
#include "common.h"

#include <stdlib.h>

#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

namespace gims {

struct OtR2Block { int prob, xr, cc, pad; };
struct OtR2Dev {
  const float* z; int64_t ld; int n, m;
  float* u; float* v; float* status;
  float norm, log_mu_bin, log_nu_bin;
  float mu, mu_bin, nu_bin;       // exp of the three: the marginals themselves (uniform values the kernel keeps in scalar registers)
  int nx, nc;          // row groups, column blocks (workgroups per row group)
  int rb, cb;          // rows per row group (<= 1024), columns per block (<= 128, multiple of 4)
  int rbf, rbs;        // row slots folded per workgroup (multiple of 4), row slots per group = nc * rbf >= rb + 1
  float* rpart;        // [nx][nc][rbs]  partial row sums (slot nrl of the last group = the dustbin row)
  float* fbuf;         // [nx][2][rbs]   F (tagged), then u on the iterations that precede a derivation
  float* cpart;        // [2][nx][nc][R2_CSEG]  partial column sums (slot 128 = the dustbin column, last block only)
  float* mpart;        // [nx][nc][rbs]  partial row maxima of Z (start potentials, exchanged once)
  int* placement;      // [1] set to 1 by any workgroup whose XCC id is not blockIdx % 8
  int* rflag;          // [iters + 8] adaptive re-derivation: rflag[s] != 0 <=> at the end of iteration s some cumulative factor had grown past its bound
};
struct OtR2Args {
  const OtR2Dev* probs;
  int p0, pcount, nx, nc;      // this launch holds problems p0 .. p0 + pcount - 1 of the class's geometry (nx row groups x nc column blocks each)
  float alpha; int iters, refresh, wt_local;      // refresh > 0: fixed period; 0: adaptive (rflag); < 0: the final derivation only
  int init_inside;     // 1: the start potentials u0 = -max(alpha, row max of Z), v0 = 0 are formed in here (no ot_init_kernel sweep of Z)
  unsigned long long* prof;
};

constexpr int R2_FLAG_ITERS = 1024;             // iterations the adaptive re-derivation has flags for (workspace: 4 KB per problem)
constexpr int R2_CSEG = 132;                    // floats per (workgroup, buffer) of the column edge: 128 columns + dustbin + pad
// LDS layout of a workgroup of R2_NT = 512 threads (a <= 1024 x 128 block, one workgroup per CU).  The K tile rows 12..15 of every thread take
// 16 R2_NT float4; the small arrays follow.  (A 256-thread geometry -- <= 512 x 128 blocks, two workgroups of two different problems per CU, so
// that one's exchange waits run under the other's sweeps -- was built in round 4, measured and removed in round 5: a lone wave per SIMD issues
// its sweep at about half the rate of two, so the two problems' chains stay as long as before: 4096 x 2 0.94 ms per launch against 0.92,
// 1024 x 32 0.87 against 0.82, whatever the start offset between the two workgroups.)
constexpr int R2_NT = 512;
struct R2L {
  static constexpr int NW = R2_NT / 64;
  static constexpr int K = 16 * R2_NT * 4;          // 4 tile rows x 4 quads per thread, float4 each
  static constexpr int NXMAX = 4;                   // row groups of a problem (rb <= 2 R2_NT rows each)
  static constexpr int ROWST = 1028, FACS = 1160, GVEC = 132, COLRED = NW * 128, PB = 2 * R2_NT + 4, PR = 132, CSST = 132;
  static constexpr int XRD = 4 * R2_CSEG;
  static constexpr int OWN = 2 * 132 + 2 * 132;     // v and G of the block's columns; u and F of the row slots this workgroup folds (rbf <= R2_FOLD = 132: r2_geom)
  static constexpr int FLOATS = K + ROWST + FACS + GVEC + COLRED + PB + PR + CSST + XRD + 16 + OWN;
};

typedef float r2f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void r2_st4_wt(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void r2_st4_plain(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void r2_ld4_issue(f32x4& v, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void r2_ld4_wait(f32x4& a) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a)::"memory"); }
__device__ __forceinline__ void r2_swap16(float& x, float& y) { asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y)); }
__device__ __forceinline__ void r2_swap32(float& x, float& y) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y)); }
template <int CTRL>
__device__ __forceinline__ float r2_dpp_max(float x) {
  return fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true)));
}
template <int CTRL>
__device__ __forceinline__ float r2_dpp_add(float x) {
  return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// all-reduce over the 8 lanes that share lane >> 3 (the 8 column groups of a tile row)
__device__ __forceinline__ float r2_sum8(float x) {
  x = r2_dpp_add<0xB1>(x);      // quad_perm [1,0,3,2]
  x = r2_dpp_add<0x4E>(x);      // quad_perm [2,3,0,1]
  return r2_dpp_add<0x141>(x);  // row_half_mirror
}

template <bool PROF>
__global__ __launch_bounds__(R2_NT, 1) void ot_res2_kernel(OtR2Args a) {
#pragma clang fp contract(off)
  using L = R2L;
  constexpr int NT = R2_NT;
  __shared__ int fail_flag;
  __shared__ unsigned long long prof_acc[8];
  unsigned long long prof_t = 0;
  auto stamp = [&](int phase) {
    if (PROF && threadIdx.x == 0) {
      const unsigned long long now = __builtin_readcyclecounter();
      if (phase >= 0) prof_acc[phase] += now - prof_t;
      prof_t = now;
    }
  };
  if (PROF && threadIdx.x < 8) prof_acc[threadIdx.x] = 0;
  if (threadIdx.x == 0) fail_flag = 0;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* klds = lds;                       // [16][NT] float4: tile rows 12..15 of every thread
  float* rowst = klds + L::K;              // row sums of this block, by row slot (slot nrl of the last group: the dustbin row)
  float* colred = rowst + L::ROWST;        // [NW waves][128] column partials
  float* facs = colred + L::COLRED;        // F of the group's row slots
  float* gvec = facs + L::FACS;            // G of the block's columns (v ahead of a derivation); [128] = the dustbin column's
  float* pb = gvec + L::GVEC;              // dustbin-column entries of the group's rows (last column block only)
  float* pr = pb + L::PB;                  // dustbin-row entries of the block's columns (last row group only)
  float* csst = pr + L::PR;                // this block's column sums, staged for the 16-byte publish
  float* xrd = csst + L::CSST;             // [nx][R2_CSEG] column partials of all row groups
  float* wred = csst + L::CSST + L::XRD;   // [16] per-wave partials of the small reductions
  // state that lives across iterations in LDS rather than in registers (the 192 registers of P leave no room): v and the
  // cumulative factor G of the block's columns (owner: thread t < 128, thread 128 the dustbin column), u and the cumulative
  // factor F of the row slots this workgroup folds (owner: the fc == 0 lane of every fold group)
  float* vown_l = wred + 16;               // [132]  v of the own columns AT THE LAST DERIVATION
  float* gown_l = vown_l + 132;            // [132]  v of the own columns now (= vown_l + log G)
  float* uo_l = gown_l + 132;              // [132]  u of the folded row slots AT THE LAST DERIVATION
  float* fo_l = uo_l + 132;                // [132]  u of the folded row slots now (= uo_l + log F)

  // workgroup -> (problem, row group, column block) by arithmetic (round 6: it was an 8-KB table uploaded per launch -- three stream-ordered upload
  // launches each): unit (problem q, row group xr) sits on XCD unit % 8 in slot unit / 8 of that XCD's workgroups, its nc column blocks on the
  // workgroups 8 (slot nc + c) + xcd -- block b runs on XCD b % 8 (checked below; speed only)
  OtR2Block bk;
  {
    const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3, slot = j / a.nc, unit = slot * 8 + xcd, q = unit / a.nx;
    bk.prob = q < a.pcount ? a.p0 + q : -1;
    bk.xr = unit - q * a.nx;
    bk.cc = j - slot * a.nc;
    bk.pad = 0;
  }
  if (bk.prob < 0) return;
  const OtR2Dev p = a.probs[bk.prob];
  const int xr = bk.xr, cc = bk.cc;
  const float alpha = a.alpha;
  const int row0 = xr * p.rb;
  int nrl = p.n - row0;
  nrl = nrl < p.rb ? nrl : p.rb;
  nrl = nrl > 0 ? nrl : 0;                                 // real rows of the group
  const bool lastg = xr == p.nx - 1;
  const int nslots = nrl + (lastg ? 1 : 0);               // + the dustbin row
  const int col0 = cc * p.cb;
  int ncl = p.m - col0;
  ncl = ncl < p.cb ? ncl : p.cb;
  ncl = ncl > 0 ? ncl : 0;                                 // real columns of the block
  const bool lastc = cc == p.nc - 1;
  const bool wt = a.wt_local != 0;
  {  // placement check (speed only: see the header)
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0 && (xcc & 7u) != (blockIdx.x & 7u)) atomicOr(p.placement, 1);
  }
  float* rpart_mine = p.rpart + (int64_t)(xr * p.nc + cc) * p.rbs;
  const float* rpart_grp = p.rpart + (int64_t)(xr * p.nc) * p.rbs;
  float* fb = p.fbuf + (int64_t)xr * 2 * p.rbs;

  // K tile of this thread: rows rg*16 + i (i < 16), columns 32 k + 4 cg + e (k < 4, e < 4); rows i < 12 in registers
  r2f2 P[12][8];
  // fold duty: thread t reads source block fc = t % nc for quad fq = t / nc of this workgroup's rbf slots (first slot fs0)
  const int nc_sh = 31 - __builtin_clz((unsigned)p.nc);    // nc is a power of two
  {
    const int fc = threadIdx.x & (p.nc - 1), fq = threadIdx.x >> nc_sh, fs0 = cc * p.rbf + 4 * fq;
    if (4 * fq < p.rbf && fc == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s = fs0 + e;
        uo_l[4 * fq + e] = a.init_inside ? 0.f : (s < nrl ? p.u[row0 + s] : (s == nrl && lastg ? p.u[p.n] : 0.f));
        fo_l[4 * fq + e] = uo_l[4 * fq + e];
      }
    }
  }
  // column duty: thread t < 128 owns column col0 + t, thread 128 of the last block the dustbin column
  if (threadIdx.x < 132) { vown_l[threadIdx.x] = 0.f; gown_l[threadIdx.x] = 0.f; }
  for (int r = threadIdx.x; r < L::ROWST; r += NT) rowst[r] = 0.f;
  for (int r = threadIdx.x; r < L::PB; r += NT) pb[r] = 0.f;
  for (int r = threadIdx.x; r < L::FACS; r += NT) facs[r] = 1.f;
  for (int r = threadIdx.x; r < L::GVEC; r += NT) { gvec[r] = 0.f; pr[r] = 0.f; csst[r] = 0.f; }     // v0 = 0
  __syncthreads();

  // ---------------- start potentials formed on chip (replaces ot_init_kernel's sweep of Z): u0_i = -max(alpha, max_j Z_ij), v0 = 0.
  // Z is read ONCE into the registers / LDS that will hold K, the row maxima take the same two local hops as the row sums
  // (readiness: the buffers start as 0xFFFFFFFF, which no maximum and no potential can be), then K = exp((Z + u0) + 0) in place.
  if (a.init_inside) {
    int tq = threadIdx.x;
    asm volatile("" : "+v"(tq));
    const int t = tq, cg = t & 7, rg = t >> 3;
    const float* zb = p.z;
    const int row_hi = p.n - 1, quad_hi = (p.m - 1) & ~3;
    constexpr float NEG = -3.0e38f;
#pragma unroll
    for (int ibs = 0; ibs < 16; ibs += 4) {
      const int ib = (ibs + 12) % 16;
      f32x4 zq[4][4];
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        int gr = row0 + rg * 16 + ib + i4;
        gr = gr < row_hi ? gr : row_hi;
        const float* zr = zb + (int64_t)gr * p.ld;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          int cq = col0 + 32 * k + 4 * cg;
          cq = cq < quad_hi ? cq : quad_hi;
          zq[i4][k] = __builtin_nontemporal_load((const f32x4*)(zr + cq));
        }
      }
      float m4[4];
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const bool rin = rg * 16 + ib + i4 < nrl;
        float mx = NEG;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float z = (rin && 32 * k + 4 * cg + e < ncl) ? zq[i4][k][e] : NEG;       // outside the block: exp(NEG + u) = 0 later
            zq[i4][k][e] = z;
            mx = fmaxf(mx, z);
          }
        m4[i4] = mx;
        if (ib < 12) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            P[(ib + i4) % 12][2 * k] = r2f2{zq[i4][k][0], zq[i4][k][1]};
            P[(ib + i4) % 12][2 * k + 1] = r2f2{zq[i4][k][2], zq[i4][k][3]};
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) ((f32x4*)klds)[((ib + i4 - 12) * 4 + k) * NT + t] = zq[i4][k];
        }
      }
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) m4[i4] = r2_dpp_max<0xB1>(m4[i4]);
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) m4[i4] = r2_dpp_max<0x4E>(m4[i4]);
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) m4[i4] = r2_dpp_max<0x141>(m4[i4]);
      if (cg == 0) *(f32x4*)(rowst + rg * 16 + ib) = f32x4{m4[0], m4[1], m4[2], m4[3]};
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    float* mpart_mine = p.mpart + (int64_t)(xr * p.nc + cc) * p.rbs;
    if (4 * t < nrl) {
      const f32x4 q = *(const f32x4*)(rowst + 4 * t);
      if (wt) r2_st4_wt(mpart_mine + 4 * t, q); else r2_st4_plain(mpart_mine + 4 * t, q);
    }
    const int fc = t & (p.nc - 1), fq = t >> nc_sh, fs0 = cc * p.rbf + 4 * fq;
    if (4 * fq < p.rbf) {
      unsigned vmask = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) vmask |= (fs0 + e < nrl ? 1u : 0u) << e;
      f32x4 q = {NEG, NEG, NEG, NEG};
      if (vmask) {
        const float* src = p.mpart + (int64_t)(xr * p.nc + fc) * p.rbs + fs0;
        int spins = 0;
        for (;;) {
          r2_ld4_issue(q, src);
          r2_ld4_wait(q);
          unsigned stale = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) stale |= (__float_as_uint(q[e]) == 0xFFFFFFFFu ? 1u : 0u) << e;
          if ((stale & vmask) == 0) break;
          if (++spins > (1 << 16)) { fail_flag = 1; atomicMax(p.placement + 1, 1); break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      f32x4 mx;
#pragma unroll
      for (int e = 0; e < 4; ++e) mx[e] = (vmask >> e) & 1u ? q[e] : NEG;
      if (p.nc >= 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) mx[e] = r2_dpp_max<0xB1>(mx[e]);
      }
      if (p.nc >= 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) mx[e] = r2_dpp_max<0x4E>(mx[e]);
      }
      if (p.nc >= 8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) mx[e] = r2_dpp_max<0x141>(mx[e]);
      }
      if (p.nc >= 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) mx[e] = r2_dpp_max<0x140>(mx[e]);
      }
      if (p.nc >= 32) {
#pragma unroll
        for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], __shfl_xor(mx[e], 16, 64));
      }
      if (fc == 0 && fs0 < nslots) {
        f32x4 u0;
#pragma unroll
        for (int e = 0; e < 4; ++e) u0[e] = fs0 + e < nrl ? -fmaxf(alpha, mx[e]) : -alpha;      // (slot nrl of the last group: the dustbin row)
        *(f32x4*)(uo_l + 4 * fq) = u0;
        *(f32x4*)(fo_l + 4 * fq) = u0;
        if (wt) r2_st4_wt(fb + p.rbs + fs0, u0); else r2_st4_plain(fb + p.rbs + fs0, u0);
      }
    }
    if (4 * t < nslots) {                      // gather the group's u0 (into facs, which is idle until the first row fold)
      f32x4 q;
      unsigned vmask = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) vmask |= (4 * t + e < nslots ? 1u : 0u) << e;
      int spins = 0;
      for (;;) {
        r2_ld4_issue(q, fb + p.rbs + 4 * t);
        r2_ld4_wait(q);
        unsigned stale = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) stale |= (__float_as_uint(q[e]) == 0xFFFFFFFFu ? 1u : 0u) << e;
        if ((stale & vmask) == 0) break;
        if (++spins > (1 << 16)) { fail_flag = 1; atomicMax(p.placement + 1, 2); break; }
        __builtin_amdgcn_s_sleep(1);
      }
      *(f32x4*)(facs + 4 * t) = q;
    }
    __syncthreads();
    // K = exp((Z + u0) + v0), v0 = 0, in place
#pragma unroll
    for (int ib = 0; ib < 16; ib += 4) {
      const f32x4 u4 = *(const f32x4*)(facs + rg * 16 + ib);
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const int i = ib + i4;
        if (i < 12) {
#pragma unroll
          for (int h = 0; h < 8; ++h) P[i % 12][h] = r2f2{__expf((P[i % 12][h][0] + u4[i4]) + 0.f), __expf((P[i % 12][h][1] + u4[i4]) + 0.f)};
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            f32x4 q = ((const f32x4*)klds)[((i - 12) * 4 + k) * NT + t];
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = __expf((q[e] + u4[i4]) + 0.f);
            ((f32x4*)klds)[((i - 12) * 4 + k) * NT + t] = q;
          }
        }
      }
    }
    if (lastc)
      for (int r = t; r < nslots; r += NT) pb[r] = __expf((alpha + facs[r]) + 0.f);       // (slot nrl: the corner)
    if (lastg && t < 128) pr[t] = t < ncl ? __expf((alpha + facs[nrl]) + 0.f) : 0.f;
    __syncthreads();
    for (int r = t; r < L::FACS; r += NT) facs[r] = 1.f;
    for (int r = t; r < L::GVEC; r += NT) gvec[r] = 1.f;
    __syncthreads();
  }

  __shared__ int nf_lds;                                     // the candidate's verdict (adaptive re-derivation), workgroup-uniform
  int last_fresh = 0;                                        // iteration of the last derivation (the start potentials count as one)
  bool pend_fresh = false;                                   // next_fresh of the previous iteration
  constexpr float R2_F_BOUND = 7.9e13f, R2_G_BOUND = 4.85e8f;      // e^32, e^20 (see the header)
  // A row or column total this small means the sum is made of entries near (or below) f32's normal range -- K entries that survived the
  // derivation as denormals carry a few bits -- so the factors it yields are wrong although finite: the solve gives up (status 2) and the
  // caller's rescue re-solves in the log domain.  Found with the reference golden raree2e_*_g10 (round 6): one keypoint with a ten times
  // larger descriptor puts |Z| at 2400 in its column, every row's maximum sits there, all other entries of K = exp(Z - rowmax) are e^-150, and
  // the solve went on with column totals of e^-100: potentials off by 4e-3, no guard tripped.  Dense and sparse fixtures stay above e^-55.
  constexpr float R2_TOT_MIN = 1.0e-30f;
  int nslots_s = __builtin_amdgcn_readfirstlane(nslots);
  const bool force_last = a.refresh > 0 || a.refresh == -1;      // fixed period / "final only": derive on the last iteration whatever happened
  for (int it = 0; it < a.iters; ++it) {
    // thread-dependent indices are re-derived from an opaque copy of the thread id every iteration (as loop invariants the
    // compiler keeps their hoisted addresses alive next to the 192 registers of P)
    int tq = threadIdx.x;
    asm volatile("" : "+v"(tq));
    // the uniform slot count lives in a SCALAR register inside the loop: as a (spilled) vector register it was reloaded from scratch in front of
    // five compares per iteration, each on the critical path of an exchange
    asm volatile("" : "+s"(nslots_s));
    const int t = tq, lane = t & 63, wave = t >> 6, cg = t & 7, rg = t >> 3;
    const bool fresh = (it == 0 && !a.init_inside) || (it > 0 && pend_fresh);
    bool next_fresh = it + 1 < a.iters && ((force_last && it + 2 == a.iters) || (a.refresh > 0 && (it + 1) % a.refresh == 0));
    if (fresh) last_fresh = it;
    if (a.refresh == 0 && (((it + 1) & 3) == 0 || it + 2 == a.iters) && it + 1 >= 8 && it + 1 < a.iters) {
      // candidate c = it + 1: flags of the iterations c - 7 .. c - 4 that lie behind the last derivation (uniform in the workgroup and, by the
      // argument in the header, in the problem)
      const int c = it + 1;
      if (t < 64) {
        const int sidx = c - 4 - (t & 3);
        const int f = (t < 4 && sidx > last_fresh) ? __hip_atomic_load(p.rflag + sidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        const unsigned long long any = __ballot(f != 0);
        if (t == 0) nf_lds = any != 0ull ? 1 : 0;
      }
      __syncthreads();
      next_fresh = nf_lds != 0;
      __syncthreads();                                       // (nf_lds is rewritten four iterations later at the earliest; this keeps the read and that write apart without relying on it)
    }
    pend_fresh = next_fresh;
    const bool need_u = next_fresh || it + 1 == a.iters;       // the potentials themselves are wanted (workgroup-uniform)
    const unsigned tagbit = (unsigned)(it & 1) << 31;
    const unsigned xtagbit = (unsigned)((it >> 1) & 1) << 31;
    auto tg = [&](float x) { return __uint_as_float(__float_as_uint(x) | tagbit); };
    stamp(-1);
    // ---------------- derivation: K = exp((Z + u) + v) from scratch
    if (fresh) {
      const float* usrc = it == 0 ? p.u + row0 : fb + p.rbs;          // u by row slot (iteration 0: by global row)
      const float u_bin_row = lastg ? (it == 0 ? p.u[p.n] : __hip_atomic_load(usrc + nrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.f;
      const float* zb = p.z;
      int row0_o = row0, col0_o = col0;
      asm volatile("" : "+s"(row0_o), "+s"(col0_o), "+s"(zb));
      // every load is unconditional with a clamped address (a select behind each load costs an exec-mask save per load: 64 of them
      // spilled the scalar file); what lies outside the block is masked when the exponentials are formed
      const int row_hi = p.n - 1, quad_hi = (p.m - 1) & ~3, u_hi = it == 0 ? p.n - row0 : p.rbs - 1;
      // tile rows two at a time (8 x 16-byte loads in flight per thread), the four LDS rows FIRST: the registers of P fill up as
      // the loop advances, so the staging registers of the last steps sit next to 160, not 192, live registers of P
      constexpr int DB = 2;
#pragma unroll
      for (int ibs = 0; ibs < 16; ibs += DB) {
        const int ib = (ibs + 12) % 16;
        f32x4 zq[DB][4];
        float ur[DB];
#pragma unroll
        for (int i4 = 0; i4 < DB; ++i4) {
          const int rl = rg * 16 + ib + i4;
          ur[i4] = __hip_atomic_load(usrc + (rl < u_hi ? rl : u_hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          int gr = row0_o + rl;
          gr = gr < row_hi ? gr : row_hi;
          const float* zr = zb + (int64_t)gr * p.ld;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            int cq = col0_o + 32 * k + 4 * cg;
            cq = cq < quad_hi ? cq : quad_hi;                        // rows are padded to 4 floats: the last quad of a row is readable
            zq[i4][k] = __builtin_nontemporal_load((const f32x4*)(zr + cq));
          }
        }
#pragma unroll
        for (int i4 = 0; i4 < DB; ++i4) {
          const int rl = rg * 16 + ib + i4;
          const bool rin = rl < nrl;
          f32x4 kq[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const f32x4 vq = *(const f32x4*)(gvec + 32 * k + 4 * cg);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float x = __expf((zq[i4][k][e] + ur[i4]) + vq[e]);
              kq[k][e] = (rin && 32 * k + 4 * cg + e < ncl) ? x : 0.f;
            }
          }
          if (ib < 12) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              P[(ib + i4) % 12][2 * k] = r2f2{kq[k][0], kq[k][1]};
              P[(ib + i4) % 12][2 * k + 1] = r2f2{kq[k][2], kq[k][3]};
            }
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) ((f32x4*)klds)[((ib + i4 - 12) * 4 + k) * NT + t] = kq[k];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // borders: dustbin column (last block) by row, dustbin row (last group) by column, corner where both
      const float vbin = gvec[128];
      if (lastc) {
        for (int r = t; r < nrl; r += NT) pb[r] = __expf((alpha + __hip_atomic_load(usrc + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + vbin);
        if (lastg && t == 0) pb[nrl] = __expf((alpha + u_bin_row) + vbin);
      }
      if (lastg && t < 128) pr[t] = t < ncl ? __expf((alpha + u_bin_row) + gvec[t]) : 0.f;
      __syncthreads();
      for (int r = t; r < L::FACS; r += NT) facs[r] = 1.f;
      for (int r = t; r < L::GVEC; r += NT) gvec[r] = 1.f;
      __syncthreads();
    }
    stamp(0);
    // ---------------- row pass: sum_j K_ij G_j over the block's columns
    {
      // the dustbin row of the last group, sum_j pr_j G_j (its corner entry is added at the publish like every row's dustbin-column entry): wave 0
      // forms it FIRST, into a word of its own that the publisher puts into slot nrl -- behind the sweep it cost the last group's workgroups a
      // barrier, a wave reduction and another barrier (~800 cycles) on the path every other workgroup of the problem waits for (round 6; formed
      // branch-free by every wave so that the scheduler could thread it through the sweep: 600 cycles MORE for every workgroup)
      if (lastg && wave == 0) {
        float s = pr[lane] * gvec[lane] + pr[lane + 64] * gvec[lane + 64];
        s = wave_sum(s);
        if (lane == 0) wred[8] = s;
      }
      r2f2 g2[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 q = *(const f32x4*)(gvec + 32 * k + 4 * cg);
        g2[2 * k] = r2f2{q[0], q[1]};
        g2[2 * k + 1] = r2f2{q[2], q[3]};
      }
      // four tile rows at a time: four independent fma chains, their 8-lane reductions interleaved, one 16-byte LDS store
      // (the dustbin column's share, pb_i G_bin, is added when the sums are published)
#pragma unroll
      for (int ib = 0; ib < 16; ib += 4) {
        float s4[4];
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          const int i = ib + i4;
          r2f2 s2;
          if (i < 12) {
            s2 = P[i % 12][0] * g2[0];
#pragma unroll
            for (int h = 1; h < 8; ++h) s2 = __builtin_elementwise_fma(P[i % 12][h], g2[h], s2);
          } else {
            s2 = r2f2{0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const f32x4 q = ((const f32x4*)klds)[((i - 12) * 4 + k) * NT + t];
              s2 = __builtin_elementwise_fma(r2f2{q[0], q[1]}, g2[2 * k], s2);
              s2 = __builtin_elementwise_fma(r2f2{q[2], q[3]}, g2[2 * k + 1], s2);
            }
          }
          s4[i4] = s2[0] + s2[1];
        }
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) s4[i4] = r2_dpp_add<0xB1>(s4[i4]);
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) s4[i4] = r2_dpp_add<0x4E>(s4[i4]);
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) s4[i4] = r2_dpp_add<0x141>(s4[i4]);
        if (cg == 0) *(f32x4*)(rowst + rg * 16 + ib) = f32x4{s4[0], s4[1], s4[2], s4[3]};
      }
    }
    __syncthreads();
    if (fail_flag) {
      if (threadIdx.x == 0) ot_raise_status(p.status, 2.f);
      return;
    }
    stamp(1);
    // ---------------- publish the row partials (16-byte stores; stay in this XCD's L2 unless wt)
    if (4 * t < nslots_s) {
      f32x4 q = *(const f32x4*)(rowst + 4 * t);
      if (lastg) {
        const float rbin = wred[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = 4 * t + e == nrl ? rbin : q[e];
      }
      if (lastc) {
        const f32x4 b4 = *(const f32x4*)(pb + 4 * t);
        const float gbin = gvec[128];
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] += b4[e] * gbin;
      }
      const f32x4 w = {tg(q[0]), tg(q[1]), tg(q[2]), tg(q[3])};
      if (wt) r2_st4_wt(rpart_mine + 4 * t, w); else r2_st4_plain(rpart_mine + 4 * t, w);
    }
    // ---------------- fold this workgroup's share of the row slots over the nc blocks; u += log f, F *= f
    // (the fold indices are re-derived from the opaque thread id here: kept across the passes above they are spilled to scratch
    // and a scratch reload costs more than the fold itself)
    const int fc = t & (p.nc - 1), fq = t >> nc_sh, fs0 = cc * p.rbf + 4 * fq;
    const bool folder = 4 * fq < p.rbf;
    if (folder) {
      f32x4 q = {0.f, 0.f, 0.f, 0.f};
      unsigned vmask = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) vmask |= (fs0 + e < nslots_s ? 1u : 0u) << e;
      if (vmask) {
        const float* src = rpart_grp + (int64_t)fc * p.rbs + fs0;
        int spins = 0;
        for (;;) {
          r2_ld4_issue(q, src);
          r2_ld4_wait(q);
          unsigned stale = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) stale |= ((__float_as_uint(q[e]) ^ tagbit) >> 31) << e;
          if ((stale & vmask) == 0) break;
          if (++spins > (1 << 16)) { fail_flag = 1; atomicMax(p.placement + 1, 3); break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      stamp(2);
      // butterfly over the nc source lanes (the same total, in the same order, in every lane): DPP inside a 16-lane row, one
      // cross-row exchange for nc = 32
      f32x4 tot;
#pragma unroll
      for (int e = 0; e < 4; ++e) tot[e] = (vmask >> e) & 1u ? fabsf(q[e]) : 0.f;
      if (p.nc >= 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) tot[e] = r2_dpp_add<0xB1>(tot[e]);
      }
      if (p.nc >= 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) tot[e] = r2_dpp_add<0x4E>(tot[e]);
      }
      if (p.nc >= 8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) tot[e] = r2_dpp_add<0x141>(tot[e]);
      }
      if (p.nc >= 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) tot[e] = r2_dpp_add<0x140>(tot[e]);
      }
      if (p.nc >= 32) {
#pragma unroll
        for (int e = 0; e < 4; ++e) tot[e] += __shfl_xor(tot[e], 16, 64);
      }
      if (fc == 0 && vmask) {
        // F_i = mu_i / sum_j K_ij G_j and u_i = u_i(last derivation) + log F_i, both from THIS iteration's sum alone: nothing is accumulated
        // from iteration to iteration (u += du at |u| ~ 100 lost half an ulp, 4e-6, per iteration against F *= exp(du): 4e-5 on the
        // marginals after 100 iterations unless K was re-derived from u at the end).  Branch-free over the four slots: this is the middle of
        // the iteration's longest dependency chain.
        // A sum outside f32's range: the cumulative factors F, G of this multiplicative form ran out of range (or the input is not finite).
        // Status 2 = "gave up": the caller's rescue re-solves the problem with the log-domain kernels, which decide whether the marginals
        // themselves are finite (status 1) -- the lazy factors never cost a pair its matches
        bool fgrow = false, bad = false;
        f32x4 fw;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int s = fs0 + e;
          const bool val = s < nslots_s, real = s < nrl;
          const float tt = val ? tot[e] : 1.f;
          const float fn = val ? (real ? p.mu : p.mu_bin) * __builtin_amdgcn_rcpf(tt) : 1.f;
          bad |= !(tt > R2_TOT_MIN) || !(tt < 3.0e38f) || !(fn < 3.0e38f);
          fgrow |= fn > R2_F_BOUND;
          fw[e] = tg(fn);
        }
        if (bad) ot_raise_status(p.status, 2.f);
        // u itself is only read by the next derivation and by the output: since nothing is accumulated (u = u(last derivation) + log F of THIS
        // iteration), its four logarithms leave the critical path of every other iteration (round 6: ~800 of the fold's 1840 cycles)
        f32x4 uo = *(const f32x4*)(uo_l + 4 * fq);
        if (need_u) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int s = fs0 + e;
            const bool val = s < nslots_s, real = s < nrl;
            uo[e] += val ? (real ? p.norm : p.log_mu_bin) - logf(tot[e]) : 0.f;
          }
          *(f32x4*)(fo_l + 4 * fq) = uo;
          if (next_fresh) *(f32x4*)(uo_l + 4 * fq) = uo;
        }
        if (a.refresh == 0 && fgrow) {                                       // (before this iteration's publishes of this lane: they order it for the readers)
          __hip_atomic_store(p.rflag + it, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __threadfence();
        }
        if (next_fresh) {                                                    // u first, acknowledged, then the tagged F readers wait on
          if (wt) r2_st4_wt(fb + p.rbs + fs0, uo); else r2_st4_plain(fb + p.rbs + fs0, uo);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (wt) r2_st4_wt(fb + fs0, fw); else r2_st4_plain(fb + fs0, fw);
      }
    }
    stamp(3);
    // ---------------- gather the group's F
    if (4 * t < nslots_s) {
      f32x4 q;
      unsigned vmask = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) vmask |= (4 * t + e < nslots_s ? 1u : 0u) << e;
      int spins = 0;
      for (;;) {
        r2_ld4_issue(q, fb + 4 * t);
        r2_ld4_wait(q);
        unsigned stale = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) stale |= ((__float_as_uint(q[e]) ^ tagbit) >> 31) << e;
        if ((stale & vmask) == 0) break;
        if (++spins > (1 << 16)) { fail_flag = 1; atomicMax(p.placement + 1, 4); break; }
        __builtin_amdgcn_s_sleep(1);
      }
      *(f32x4*)(facs + 4 * t) = f32x4{fabsf(q[0]), fabsf(q[1]), fabsf(q[2]), fabsf(q[3])};
    }
    __syncthreads();
    stamp(4);
    // ---------------- column pass: sum_i F_i K_ij over the group's rows, for the block's columns
    {
      r2f2 cs2[8];
#pragma unroll
      for (int h = 0; h < 8; ++h) cs2[h] = r2f2{0.f, 0.f};
#pragma unroll
      for (int ib = 0; ib < 16; ib += 4) {
        const f32x4 f4 = *(const f32x4*)(facs + rg * 16 + ib);
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          const int i = ib + i4;
          const r2f2 f2 = {f4[i4], f4[i4]};
          if (i < 12) {
#pragma unroll
            for (int h = 0; h < 8; ++h) cs2[h] = __builtin_elementwise_fma(P[i % 12][h], f2, cs2[h]);
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const f32x4 q = ((const f32x4*)klds)[((i - 12) * 4 + k) * NT + t];
              cs2[2 * k] = __builtin_elementwise_fma(r2f2{q[0], q[1]}, f2, cs2[2 * k]);
              cs2[2 * k + 1] = __builtin_elementwise_fma(r2f2{q[2], q[3]}, f2, cs2[2 * k + 1]);
            }
          }
        }
      }
      // reduce over the 8 tile rows of the wave (lane bits 3, 4, 5): value 4 k + e <-> column 32 k + 4 cg + e
      float v[16];
#pragma unroll
      for (int h = 0; h < 8; ++h) { v[2 * h] = cs2[h][0]; v[2 * h + 1] = cs2[h][1]; }
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = r2_dpp_add<0x128>(v[i]);           // row_ror:8 -- lane ^ 8
      float w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { r2_swap16(v[2 * j], v[2 * j + 1]); w[j] = v[2 * j] + v[2 * j + 1]; }     // 16-lane rows 0,2: value 2j; rows 1,3: value 2j+1
      float z4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { r2_swap32(w[2 * j], w[2 * j + 1]); z4[j] = w[2 * j] + w[2 * j + 1]; }   // 16-lane row q: value 4j + q
      if ((lane & 8) == 0) {
        const int q = lane >> 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) colred[wave * 128 + 32 * j + 4 * cg + q] = z4[j];
      }
      if (lastc) {              // dustbin column: sum_i F_i pb_i over the group's slots (the corner included: it sits in slot nrl)
        float s = 0.f;
        for (int r = t; r < nslots_s; r += NT) s += facs[r] * pb[r];
        s = wave_sum(s);
        if (lane == 0) wred[wave] = s;
      }
    }
    __syncthreads();
    float ctot = 0.f;                                                          // this row group's partial for the thread's column
    if (t < 128) {
#pragma unroll
      for (int w8 = 0; w8 < L::NW; ++w8) ctot += colred[w8 * 128 + t];
      if (lastg) ctot += facs[nrl] * pr[t];
      csst[t] = ctot;
    } else if (t == 128 && lastc) {
#pragma unroll
      for (int w8 = 0; w8 < L::NW; ++w8) ctot += wred[w8];
      csst[128] = ctot;
    }
    stamp(5);
    // ---------------- column edge: the nx workgroups that hold the same columns exchange their partials (write-through: cross-XCD)
    if (p.nx > 1) {
      __syncthreads();
      float* cmine = p.cpart + ((int64_t)(((it & 1) * p.nx + xr) * p.nc + cc)) * R2_CSEG;
      const auto xtg = [&](float x) { return __uint_as_float(__float_as_uint(x) | xtagbit); };
      if (t < 33) {
        const f32x4 q = *(const f32x4*)(csst + 4 * t);
        r2_st4_wt(cmine + 4 * t, f32x4{xtg(q[0]), xtg(q[1]), xtg(q[2]), xtg(q[3])});
        *(f32x4*)(xrd + xr * R2_CSEG + 4 * t) = q;
      }
      // one (other row group, quad) pair per thread, from thread 64 on (the publishing wave does not also wait); with eight groups and 256
      // threads the pairs wrap around once
      for (int idx = t - 64 < 0 ? t - 64 + NT : t - 64; idx < 33 * (p.nx - 1); idx += NT) {
        const int o = idx / 33, qd = idx % 33;
        const int xs = o < xr ? o : o + 1;                                    // the other row groups, in order
        const float* src = p.cpart + ((int64_t)(((it & 1) * p.nx + xs) * p.nc + cc)) * R2_CSEG + 4 * qd;
        unsigned vmask = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int c = 4 * qd + e; vmask |= ((c < ncl || (c == 128 && lastc)) ? 1u : 0u) << e; }
        f32x4 q = {0.f, 0.f, 0.f, 0.f};
        int spins = 0;
        if (vmask)
          for (;;) {
            r2_ld4_issue(q, src);
            r2_ld4_wait(q);
            unsigned stale = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) stale |= ((__float_as_uint(q[e]) ^ xtagbit) >> 31) << e;
            if ((stale & vmask) == 0) break;
            if (++spins > (1 << 16)) { fail_flag = 1; atomicMax(p.placement + 1, 5); break; }
            __builtin_amdgcn_s_sleep(1);
          }
        *(f32x4*)(xrd + xs * R2_CSEG + 4 * qd) = f32x4{fabsf(q[0]), fabsf(q[1]), fabsf(q[2]), fabsf(q[3])};
      }
      __syncthreads();
      if (t <= 128) {
        ctot = 0.f;
        for (int xs = 0; xs < p.nx; ++xs) ctot += xrd[xs * R2_CSEG + t];       // fixed order: the same bits in all nx workgroups
      }
    }
    stamp(6);
    // ---------------- v += log g, G *= g for the block's own columns
    if (t <= 128) {
      const bool own = t < ncl || (t == 128 && lastc);
      float vown = vown_l[t], gown = 1.f;
      if (own) {
        gown = (t < 128 ? p.mu : p.nu_bin) * __builtin_amdgcn_rcpf(ctot);       // G_j = nu_j / sum_i F_i K_ij (see the row update)
        if (!(ctot > R2_TOT_MIN) || !(ctot < 3.0e38f) || !(gown < 3.0e38f)) ot_raise_status(p.status, 2.f);
        if (need_u) {                                                            // (v like u: only ahead of a derivation and on the last iteration)
          vown += (t < 128 ? p.norm : p.log_nu_bin) - logf(ctot);
          gown_l[t] = vown;
          if (next_fresh) vown_l[t] = vown;
        }
        if (a.refresh == 0 && gown > R2_G_BOUND) {
          __hip_atomic_store(p.rflag + it, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __threadfence();
        }
      }
      gvec[t] = own ? (next_fresh ? vown : gown) : (next_fresh ? 0.f : 1.f);
    }
    __syncthreads();
    stamp(7);
  }
  if (PROF && blockIdx.x == 0 && threadIdx.x < 8) a.prof[threadIdx.x] = prof_acc[threadIdx.x];
  __syncthreads();
  if (fail_flag) {
    if (threadIdx.x == 0) ot_raise_status(p.status, 2.f);
    return;
  }
  // ---------------- potentials out (the selection kernels read Z, u, v)
  {
    const int fc = threadIdx.x & (p.nc - 1), fq = threadIdx.x >> nc_sh, fs0 = cc * p.rbf + 4 * fq;
    if (4 * fq < p.rbf && fc == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s = fs0 + e;
        if (s < nrl) p.u[row0 + s] = fo_l[4 * fq + e];
        else if (s == nrl && lastg) p.u[p.n] = fo_l[4 * fq + e];
      }
    }
  }
  if (xr == 0) {
    if ((int)threadIdx.x < ncl) p.v[col0 + threadIdx.x] = gown_l[threadIdx.x];
    else if (threadIdx.x == 128 && lastc) p.v[p.m] = gown_l[128];
  }
}

// ------------------------------------------------------------------------------------------------ host side
static int r2_env(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}
static inline size_t r2_al(size_t x) { return (x + 255) & ~(size_t)255; }
static inline int r2_up4(int x) { return (x + 3) & ~3; }

// Geometry classes.  The decomposition of a problem -- nx row groups x nc column blocks, hence every summation order inside the solve -- is a
// function of ITS OWN size only: a ragged list is split into classes of equal (nx, nc) and every class gets its own launches.  (One geometry
// per call, taken from the largest problem, made a 256-keypoint pair's potentials depend on what else was in the batch: 5.6e-6 on the scores
// between match_pairs and forward() on the same pair.)
// (nx, nc) of one problem: nx row groups of <= 1024 rows; nc = the smallest power of two of <= 128-column blocks that covers m, RAISED
// until the row slots a workgroup folds (rbf = up4(ceil((rb + 1) / nc)): u and F of them live in uo_l / fo_l) fit R2_FOLD = 132 --
// a tall, narrow problem (m <= 512 with more than ~132 nc rows per group) otherwise indexes past those arrays.  nc = 32 always fits
// (rbf <= 36).  false: no on-chip geometry (empty or oversized problem).
constexpr int R2_FOLD = 132;
static inline int r2_rbmax() { return 1024; }
static inline bool r2_geom(int n, int m, int& nx, int& nc) {
  if (n < 1 || m < 1 || n > 4096 || m > 4096) return false;
  nc = 1;
  while (nc * 128 < m) nc *= 2;
  nx = cdiv(n, r2_rbmax());
  const int rb = cdiv(n, nx);
  while (nc < 32 && r2_up4(cdiv(rb + 1, nc)) > R2_FOLD) nc *= 2;
  return r2_up4(cdiv(rb + 1, nc)) <= R2_FOLD;
}
static inline int r2_class_key(const OtR2Host& h) {
  int nx = 0, nc = 0;
  return r2_geom(h.n, h.m, nx, nc) ? nx * 64 + nc : -1;
}

static OtR2Plan plan_class(const OtR2Host* pr, int np, int iters) {
  OtR2Plan P{};
  if (iters < 1 || np < 1) return P;
  int nx = 0, nc = 0;
  if (!r2_geom(pr[0].n, pr[0].m, nx, nc)) return P;                 // (every problem of a class has the same geometry: r2_classes)
  for (int i = 1; i < np; ++i) {
    int nxi = 0, nci = 0;
    if (!r2_geom(pr[i].n, pr[i].m, nxi, nci) || nxi != nx || nci != nc) return P;
  }
  const int units = 8 * (32 / nc);                                  // row groups one launch holds (each on the CUs of one XCD)
  if (nx > units) return P;
  P.nx = nx; P.nc = nc;
  P.ppg = units / nx;
  P.ppg = P.ppg < np ? P.ppg : np;
  P.ngroups = cdiv(np, P.ppg);
  if (P.ngroups > 64) return P;
  size_t b = r2_al(sizeof(OtR2Dev) * (size_t)np) + (size_t)P.ngroups * r2_al(sizeof(OtR2Block) * 512) + 256;
  for (int i = 0; i < np; ++i) {
    const int rb = cdiv(pr[i].n, nx);
    const int rbf = r2_up4(cdiv(rb + 1, nc)), rbs = nc * rbf;
    if (rbf > R2_FOLD || rb > r2_rbmax() || r2_up4(cdiv(pr[i].m, nc)) > 128) return P;
    b += 2 * r2_al((size_t)nx * nc * rbs * 4) + r2_al((size_t)nx * 2 * rbs * 4) + r2_al((size_t)2 * nx * nc * R2_CSEG * 4);
    b += r2_al((size_t)(R2_FLAG_ITERS + 8) * 4);                       // rflag: a fixed capacity -- the caller's workspace query does not know the iteration count
  }
  P.bytes = b;
  P.ok = true;
  return P;
}

static std::vector<std::vector<int>> r2_classes(const OtR2Host* pr, int np) {
  std::vector<std::vector<int>> out;
  std::vector<int> keys;
  for (int i = 0; i < np; ++i) {
    const int k = r2_class_key(pr[i]);
    size_t c = 0;
    while (c < keys.size() && keys[c] != k) ++c;
    if (c == keys.size()) { keys.push_back(k); out.emplace_back(); }
    out[c].push_back(i);
  }
  return out;
}

OtR2Plan ot_res2_plan(const OtR2Host* pr, int np, int iters) {
  OtR2Plan P{};
  if (iters < 1 || np < 1) return P;
  P.ok = true;
  for (const std::vector<int>& cls : r2_classes(pr, np)) {
    std::vector<OtR2Host> sub;
    for (int i : cls) sub.push_back(pr[i]);
    const OtR2Plan c = plan_class(sub.data(), (int)sub.size(), iters);
    if (!c.ok) return OtR2Plan{};
    P.nx = c.nx > P.nx ? c.nx : P.nx;
    P.nc = c.nc > P.nc ? c.nc : P.nc;
    P.ppg = c.ppg > P.ppg ? c.ppg : P.ppg;
    P.ngroups += c.ngroups;
    P.bytes += r2_al(c.bytes);
  }
  return P;
}

static int run_class(const OtR2Plan& P, const OtR2Host* hp, int np, float alpha, int iters, int init_inside, char* base, hipStream_t s);

// per-device state of the launcher (created once per device under a lock)
struct R2State { int resident_ok, wt_local; volatile int* h_place; hipEvent_t place_ev; };
static R2State* r2_state(const void* kernel, int threads, int blocks, size_t lds) {
  static std::mutex mu;
  static std::map<int, R2State> states;
  const int dev = current_device();
  std::lock_guard<std::mutex> lock(mu);
  auto it = states.find(dev);
  if (it != states.end()) return &it->second;
  R2State st{};
  int per_cu = 0, cus = 0;
  const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
  st.resident_ok = (e == hipSuccess && per_cu * cus >= blocks) ? 1 : 0;
  st.wt_local = r2_env("GIMS_OT_R2_WT", 0) ? 1 : 0;
  st.h_place = (volatile int*)pinned_once("ot_res2_place", 256);
  if (!st.h_place || hipEventCreateWithFlags(&st.place_ev, hipEventDisableTiming) != hipSuccess) return nullptr;
  return &states.emplace(dev, st).first->second;
}

int ot_res2_run(const OtR2Plan&, const OtR2Host* hp, int np, float alpha, int iters, int init_inside, char* base, hipStream_t s) {
  size_t off = 0;
  for (const std::vector<int>& cls : r2_classes(hp, np)) {
    std::vector<OtR2Host> sub;
    for (int i : cls) sub.push_back(hp[i]);
    const OtR2Plan c = plan_class(sub.data(), (int)sub.size(), iters);
    if (!c.ok) { set_error("ot_res2_run: a geometry class has no plan"); return GIMS_EINVAL; }
    const int rc = run_class(c, sub.data(), (int)sub.size(), alpha, iters, init_inside, base + off, s);
    if (rc != GIMS_OK) return rc;
    off += r2_al(c.bytes);
  }
  return GIMS_OK;
}

static int run_class(const OtR2Plan& P, const OtR2Host* hp, int np, float alpha, int iters, int init_inside, char* base, hipStream_t s) {
  size_t off = 0;
  OtR2Dev* dprob = (OtR2Dev*)(base + off); off += r2_al(sizeof(OtR2Dev) * (size_t)np);
  int* dplace = (int*)(base + off); off += 256;
  std::vector<OtR2Dev> hd(np);
  const size_t ex0 = off;
  for (int i = 0; i < np; ++i) {
    const OtR2Host& d = hp[i];
    OtR2Dev q{};
    q.z = d.z; q.ld = d.ld; q.n = d.n; q.m = d.m; q.u = d.u; q.v = d.v; q.status = d.status;
    q.norm = d.norm; q.log_mu_bin = d.log_mu_bin; q.log_nu_bin = d.log_nu_bin;
    q.mu = (float)exp((double)d.norm); q.mu_bin = (float)exp((double)d.log_mu_bin); q.nu_bin = (float)exp((double)d.log_nu_bin);
    q.nx = P.nx; q.nc = P.nc;
    q.rb = cdiv(d.n, P.nx);
    q.cb = r2_up4(cdiv(d.m, P.nc));
    q.rbf = r2_up4(cdiv(q.rb + 1, P.nc));
    q.rbs = P.nc * q.rbf;
    q.rpart = (float*)(base + off); off += r2_al((size_t)P.nx * P.nc * q.rbs * 4);
    q.fbuf = (float*)(base + off); off += r2_al((size_t)P.nx * 2 * q.rbs * 4);
    q.cpart = (float*)(base + off); off += r2_al((size_t)2 * P.nx * P.nc * R2_CSEG * 4);
    q.mpart = (float*)(base + off); off += r2_al((size_t)P.nx * P.nc * q.rbs * 4);
    q.placement = dplace;
    hd[i] = q;
  }
  // exchange buffers start with every sign bit set: iteration 0 waits for sign 0
  GIMS_HIP(hipMemsetAsync(base + ex0, 0xFF, off - ex0, s));
  {   // the re-derivation flags of all problems, zeroed (behind the exchange buffers)
    const size_t rf0 = off, rfb = r2_al((size_t)(R2_FLAG_ITERS + 8) * 4);
    for (int i = 0; i < np; ++i) { hd[i].rflag = (int*)(base + off); off += rfb; }
    GIMS_HIP(hipMemsetAsync(base + rf0, 0, off - rf0, s));
  }
  GIMS_HIP(hipMemsetAsync(dplace, 0, 256, s));
  int rc = upload_table(hd.data(), sizeof(OtR2Dev) * (size_t)np, dprob, s);
  if (rc != GIMS_OK) return rc;
  const int nthreads = R2_NT, nblocks = 256;
  const size_t lds = R2L::FLOATS * sizeof(float);
  GIMS_LDS_ATTR((const void*)ot_res2_kernel<false>, (int)(R2L::FLOATS * sizeof(float)));
  GIMS_LDS_ATTR((const void*)ot_res2_kernel<true>, (int)(R2L::FLOATS * sizeof(float)));
  R2State* st = r2_state((const void*)ot_res2_kernel<false>, nthreads, nblocks, lds);
  if (!st) { set_error("ot_res2_run: no per-device state"); return GIMS_EHIP; }
  // all workgroups of a launch wait on each other: they must be co-resident (one / two per CU) -- checked once per device against the occupancy query
  if (!st->resident_ok) {
    set_error("the on-chip Sinkhorn kernel does not fit: %d workgroups of %d threads with %zu bytes of LDS are not co-resident on this device", nblocks, nthreads, lds);
    return GIMS_EHIP;
  }
  // write-through stores on the local edges as well when the dispatcher was seen to place blocks elsewhere than XCD b % 8 (sticky per
  // device; read back lazily: the flag of call k is looked at by call k + 1, without a synchronisation of its own)
  if (*st->h_place) st->wt_local = 1;
  const int wt_local = st->wt_local;
  // GIMS_OT_REFRESH: k > 0 = a derivation every k iterations (rounds 2-4: 50); 0 (default) = adaptive, see the header; -1 = the final one only
  int refresh = r2_env("GIMS_OT_REFRESH", 0);
  if (refresh == 0 && iters > R2_FLAG_ITERS) refresh = 50;       // more iterations than flags: the fixed period of rounds 2-4
  const int prof = r2_env("GIMS_OT_PROF", 0);
  for (int gi = 0; gi < P.ngroups; ++gi) {
    const int p0 = gi * P.ppg, p1 = (p0 + P.ppg < np) ? p0 + P.ppg : np;
    // (unit (problem q, row group xr) -> XCD unit % 8, slot unit / 8 of that XCD's workgroups: computed in the kernel from p0, pcount, nx, nc)
    OtR2Args a{};
    a.probs = dprob; a.p0 = p0; a.pcount = p1 - p0; a.nx = P.nx; a.nc = P.nc; a.alpha = alpha; a.iters = iters; a.refresh = refresh; a.wt_local = wt_local; a.init_inside = init_inside;
    if (prof) {
      unsigned long long* dprof = (unsigned long long*)device_once("ot_res2_prof", 8 * sizeof(unsigned long long), nullptr);
      GIMS_CHECK_ARG(dprof, "ot_res2_run: no profile buffer");
      a.prof = dprof;
      hipLaunchKernelGGL((ot_res2_kernel<true>), dim3(256), dim3(R2_NT), lds, s, a);
      GIMS_LAUNCH_CHECK();
      unsigned long long h[8];
      GIMS_HIP(hipStreamSynchronize(s));
      GIMS_HIP(hipMemcpy(h, dprof, sizeof(h), hipMemcpyDeviceToHost));
      static const char* names[8] = {"derive", "row pass", "publish + wait for the row partials", "fold + publish F", "gather F (incl. wait)", "column pass",
                                     "column edge (incl. wait)", "column update"};
      fprintf(stderr, "[ot_res2 nx=%d nc=%d iters=%d wt=%d] cycles/iteration of workgroup 0:", P.nx, P.nc, iters, wt_local);
      for (int i = 0; i < 8; ++i) fprintf(stderr, "  %s %.0f;", names[i], (double)h[i] / iters);
      fprintf(stderr, "\n");
    } else {
      hipLaunchKernelGGL((ot_res2_kernel<false>), dim3(256), dim3(R2_NT), lds, s, a);
      GIMS_LAUNCH_CHECK();
    }
  }
  if (r2_env("GIMS_OT_R2_DEBUG", 0)) {          // diagnostics (synchronous): which bounded wait ran out, by site number in source order
    int h[4] = {0, 0, 0, 0};
    GIMS_HIP(hipStreamSynchronize(s));
    GIMS_HIP(hipMemcpy(h, dplace, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[ot_res2 debug] nx=%d nc=%d placement flag %d, last bounded wait that ran out: site %d\n", P.nx, P.nc, h[0], h[1]);
  }
  if (!wt_local) {
    if (hipEventQuery(st->place_ev) != hipErrorNotReady) {      // the previous read-back (if any) has landed: start the next one
      GIMS_HIP(hipMemcpyAsync((void*)st->h_place, dplace, sizeof(int), hipMemcpyDeviceToHost, s));
      GIMS_HIP(hipEventRecord(st->place_ev, s));
    }
  }
  return GIMS_OK;
}

}  // namespace gims

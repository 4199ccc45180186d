// Log-domain Sinkhorn optimal transport + mutual-argmax selection (gmatcher.py:41-69, 284-294).
//
// HBM-bound: the reference reads (and re-materialises) the (N+1)x(M+1) matrix twice per iteration.
// Here the iteration is run in the absorbed-potential form
//     P_ij = exp(Z_ij + u_i + v_j)
//     row:  r_i = sum_j P_ij ;  u_i += log mu_i - log r_i        (== log mu - LSE_j(Z + v))
//     col:  c_j = sum_i P_ij ;  v_j += log nu_j - log c_j        (== log nu - LSE_i(Z + u))
// and ONE sweep over Z serves a row update and the following column update: a workgroup owns a slab of
// 8 rows, keeps e_ij = exp(Z_ij + u_i + v_j) in registers, reduces the row sums, then accumulates the
// column sums of the *updated* matrix as e_ij * (mu_i / r_i) straight from registers.  Column sums are
// written as per-workgroup partials and folded by a tiny second kernel (fixed order -> deterministic).
// After the first row normalisation every P_ij <= 1, so no running max is needed; the start potentials
// u0_i = -max_j Z_ij make the first sweep safe as well (the recurrence does not depend on u0).
// The dustbin row / column (all = alpha) are handled analytically; Z stays the inner N x M block.
// Traffic per iteration: N*M*4 bytes of Z + 2 * G*(M+1)*4 bytes of partials (G <= 512 workgroups).
#include "common.h"

#include <string.h>

#include <vector>

namespace gims {

constexpr int OT_R = 8;  // rows per slab

struct OtDev {
  const float* z; int64_t ld; int n, m;
  float* u; float* v;            // [n+1], [m+1]
  float* partial;                // [G][m+1]
  float* cbest_val; int* cbest_idx;  // [G][m]
  float* max0; int* idx0;        // [n]
  float* max1; int* idx1;        // [m]
  int64_t* matches0; int64_t* matches1; float* mscores0; float* mscores1;
  float* status;                 // 1 word: 0 ok / 1 numeric guard
  int G;
  float norm, log_mu_bin, log_nu_bin;  // norm = -log(n+m); log(m)+norm; log(n)+norm
};

// Reduce R (= 8, 4 or 2) per-lane values over the 64 lanes of a wave with a reduce-scatter butterfly:
// log2(R) exchange steps halve the number of live values, the rest fold the single survivor -- R-1 + (6-log2 R)
// cross-lane moves instead of 6*R.  On return lane l holds the total of row
// ((l>>5)&1)*R/2 + ((l>>4)&1)*R/4 + ... ; lanes with the low (6 - log2 R) bits clear are the writers.
template <int R>
__device__ __forceinline__ float wave_reduce_rows(float (&v)[R], int lane, int& row_out) {
  float cur[R];
#pragma unroll
  for (int i = 0; i < R; ++i) cur[i] = v[i];
  int n = R, bit = 32, row = 0;
#pragma unroll
  for (; n > 1; n >>= 1, bit >>= 1) {
    const bool up = lane & bit;
    const int h = n >> 1;
#pragma unroll
    for (int i = 0; i < h; ++i) {
      const float send = up ? cur[i] : cur[i + h];
      const float mine = up ? cur[i + h] : cur[i];
      cur[i] = mine + __shfl_xor(send, bit, 64);
    }
    row += up ? h : 0;
  }
  float x = cur[0];
#pragma unroll
  for (; bit > 0; bit >>= 1) x += __shfl_xor(x, bit, 64);
  row_out = row;
  return x;
}

// ---------------------------------------------------------------------------------------------- init
__global__ void ot_init_kernel(const OtDev* __restrict__ probs, float alpha, int zero_init) {
  const OtDev p = probs[blockIdx.y];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (blockIdx.x == 0) {
    for (int j = threadIdx.x; j <= p.m; j += blockDim.x) p.v[j] = 0.f;
    if (threadIdx.x == 0) { p.u[p.n] = zero_init ? 0.f : -alpha; p.status[0] = 0.f; }
  }
  if (row >= p.n) return;
  const float* zr = p.z + (int64_t)row * p.ld;
  float mx = alpha;
  for (int j = lane; j < p.m; j += 64) mx = fmaxf(mx, zr[j]);
  mx = wave_max(mx);
  if (lane == 0) p.u[row] = zero_init ? 0.f : -mx;   // iters == 0: the reference returns Z + 0 + 0 - norm
}

// ---------------------------------------------------------------------------------------------- fused iteration
template <int CPT, int R>
__global__ __launch_bounds__(1024) void ot_iter_kernel(const OtDev* __restrict__ probs, float alpha) {
  __shared__ float red[16][R];
  __shared__ float fac[R];
  const OtDev p = probs[blockIdx.y];
  if ((int)blockIdx.x >= p.G) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = blockDim.x >> 6;

  // owned columns: quads q = t + blockDim*s
  float4 vq[CPT], acc[CPT];
#pragma unroll
  for (int s = 0; s < CPT; ++s) {
    const int c0 = 4 * (t + blockDim.x * s);
    vq[s].x = c0 + 0 < p.m ? p.v[c0 + 0] : 0.f;
    vq[s].y = c0 + 1 < p.m ? p.v[c0 + 1] : 0.f;
    vq[s].z = c0 + 2 < p.m ? p.v[c0 + 2] : 0.f;
    vq[s].w = c0 + 3 < p.m ? p.v[c0 + 3] : 0.f;
    acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float vbin = p.v[p.m];
  float accbin = 0.f;

  // ---- dustbin row (all entries alpha), handled once by workgroup 0
  if (blockIdx.x == 0) {
    const float ub = p.u[p.n];
    float4 e[CPT];
    float rs = 0.f;
#pragma unroll
    for (int s = 0; s < CPT; ++s) {
      const int c0 = 4 * (t + blockDim.x * s);
      e[s].x = c0 + 0 < p.m ? __expf(alpha + ub + vq[s].x) : 0.f;
      e[s].y = c0 + 1 < p.m ? __expf(alpha + ub + vq[s].y) : 0.f;
      e[s].z = c0 + 2 < p.m ? __expf(alpha + ub + vq[s].z) : 0.f;
      e[s].w = c0 + 3 < p.m ? __expf(alpha + ub + vq[s].w) : 0.f;
      rs += (e[s].x + e[s].y) + (e[s].z + e[s].w);
    }
    rs = wave_sum(rs);
    if (lane == 0) red[wave][0] = rs;
    __syncthreads();
    if (t == 0) {
      float tot = 0.f;
      for (int w = 0; w < nw; ++w) tot += red[w][0];
      const float corner = __expf(alpha + ub + vbin);
      tot += corner;
      if (!(tot > 0.f) || !(tot < 3.0e38f)) p.status[0] = 1.f;
      const float du = p.log_mu_bin - logf(tot);
      p.u[p.n] = ub + du;
      const float f = __expf(du);
      fac[0] = f;
      accbin += corner * f;
    }
    __syncthreads();
    const float f = fac[0];
#pragma unroll
    for (int s = 0; s < CPT; ++s) {
      acc[s].x += e[s].x * f; acc[s].y += e[s].y * f; acc[s].z += e[s].z * f; acc[s].w += e[s].w * f;
    }
    __syncthreads();
  }

  // ---- slabs of R rows.  The Z rows of the NEXT slab are requested before the current slab is reduced, so
  // the HBM stream keeps running through the barrier-separated reduce / rescale phases (one workgroup per CU).
  const int n_slabs = (p.n + R - 1) / R;
  float4 zn[R][CPT];
  float un[R];
  // branch-free loads: rows are clamped, column quads past m are redirected to column 0 and masked by value
  // (scores rows are padded to a multiple of 4 floats, so a quad that starts inside the row ends inside it)
  int cl[CPT];
  bool cm[CPT][4];
#pragma unroll
  for (int s = 0; s < CPT; ++s) {
    const int c0 = 4 * (t + blockDim.x * s);
    cl[s] = c0 < p.m ? c0 : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) cm[s][k] = c0 + k < p.m;
  }
  const float* __restrict__ zbase = p.z;
  const float* __restrict__ ubase = p.u;
  auto load_slab = [&](int slab) {
    const int r0 = slab * R;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = r0 + r < p.n ? r0 + r : p.n - 1;
      un[r] = ubase[row];
      const float* zr = zbase + (int64_t)row * p.ld;
#pragma unroll
      for (int s = 0; s < CPT; ++s) {   // streamed once per launch: non-temporal
        const f32x4 z = __builtin_nontemporal_load((const f32x4*)(zr + cl[s]));
        zn[r][s] = make_float4(z[0], z[1], z[2], z[3]);
      }
    }
  };
  if ((int)blockIdx.x < n_slabs) load_slab(blockIdx.x);
  for (int slab = blockIdx.x; slab < n_slabs; slab += p.G) {
    const int r0 = slab * R;
    float4 e[R][CPT];
    float rs[R], ucur[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      ucur[r] = un[r];
#pragma unroll
      for (int s = 0; s < CPT; ++s) e[r][s] = zn[r][s];
    }
    if (slab + p.G < n_slabs) load_slab(slab + p.G);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool live = r0 + r < p.n;
      const float ui = ucur[r];
      rs[r] = 0.f;
#pragma unroll
      for (int s = 0; s < CPT; ++s) {
        const float4 z = e[r][s];
        float4 x;
        x.x = (live && cm[s][0]) ? __expf(z.x + ui + vq[s].x) : 0.f;
        x.y = (live && cm[s][1]) ? __expf(z.y + ui + vq[s].y) : 0.f;
        x.z = (live && cm[s][2]) ? __expf(z.z + ui + vq[s].z) : 0.f;
        x.w = (live && cm[s][3]) ? __expf(z.w + ui + vq[s].w) : 0.f;
        e[r][s] = x;
        rs[r] += (x.x + x.y) + (x.z + x.w);
      }
    }
    {
      int rrow;
      const float w = wave_reduce_rows<R>(rs, lane, rrow);
      if ((lane & (64 / R - 1)) == 0) red[wave][rrow] = w;
    }
    __syncthreads();
    if (t < R) {
      const int row = r0 + t;
      float f = 0.f;
      if (row < p.n) {
        float tot = 0.f;
        for (int w = 0; w < nw; ++w) tot += red[w][t];
        const float ui = p.u[row];
        const float ebin = __expf(alpha + ui + vbin);
        tot += ebin;
        if (!(tot > 0.f) || !(tot < 3.0e38f)) p.status[0] = 1.f;
        const float du = p.norm - logf(tot);
        p.u[row] = ui + du;
        f = __expf(du);
        red[0][t] = ebin * f;   // this row's share of the dustbin column (read by thread 0 below)
      } else {
        red[0][t] = 0.f;
      }
      fac[t] = f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float f = fac[r];
#pragma unroll
      for (int s = 0; s < CPT; ++s) {
        acc[s].x += e[r][s].x * f; acc[s].y += e[r][s].y * f; acc[s].z += e[r][s].z * f; acc[s].w += e[r][s].w * f;
      }
    }
    if (t == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) accbin += red[0][r];
    }
    __syncthreads();
  }

  float* pp = p.partial + (int64_t)blockIdx.x * (p.m + 1);
#pragma unroll
  for (int s = 0; s < CPT; ++s) {
    const int c0 = 4 * (t + blockDim.x * s);
    if (c0 + 0 < p.m) pp[c0 + 0] = acc[s].x;
    if (c0 + 1 < p.m) pp[c0 + 1] = acc[s].y;
    if (c0 + 2 < p.m) pp[c0 + 2] = acc[s].z;
    if (c0 + 3 < p.m) pp[c0 + 3] = acc[s].w;
  }
  if (t == 0) pp[p.m] = accbin;
}

// v_j += log nu_j - log(sum of partials)    block = 64 columns x 16 partial groups; every thread issues its
// (<= 32) loads back to back so the fold is one memory round trip, not a dependent chain
__global__ __launch_bounds__(1024) void ot_colreduce_kernel(const OtDev* __restrict__ probs) {
  __shared__ float red[16][64];
  const OtDev p = probs[blockIdx.y];
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  if (blockIdx.x * 64 > p.m) return;
  float s = 0.f;
  if (col <= p.m) {
    const float* pp = p.partial + col;
    const int64_t st = p.m + 1;
    for (int b0 = g; b0 < p.G; b0 += 128) {   // 8 independent loads per trip
      float x[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int b = b0 + 16 * q;
        x[q] = b < p.G ? pp[(int64_t)b * st] : 0.f;
      }
      s += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
    }
  }
  red[g][cl] = s;
  __syncthreads();
  if (g == 0 && col <= p.m) {
    float c = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) c += red[q][cl];
    if (!(c > 0.f) || !(c < 3.0e38f)) p.status[0] = 1.f;
    const float lognu = col < p.m ? p.norm : p.log_nu_bin;
    p.v[col] += lognu - logf(c);
  }
}

// ---------------------------------------------------------------------------------------------- selection
// t_ij = ((Z_ij + u_i) + v_j) - norm on the inner block (gmatcher.py:47,68,284): row max/argmax directly,
// column max/argmax through per-workgroup partials (rows visited in ascending order; ties -> lower index).
template <int CPT>
__global__ __launch_bounds__(1024) void ot_select_kernel(const OtDev* __restrict__ probs) {
  __shared__ float rv[16][OT_R];
  __shared__ int ri[16][OT_R];
  const OtDev p = probs[blockIdx.y];
  if ((int)blockIdx.x >= p.G) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = blockDim.x >> 6;
  float4 vq[CPT], cb[CPT];
  int4 cbi[CPT];
#pragma unroll
  for (int s = 0; s < CPT; ++s) {
    const int c0 = 4 * (t + blockDim.x * s);
    vq[s].x = c0 + 0 < p.m ? p.v[c0 + 0] : 0.f;
    vq[s].y = c0 + 1 < p.m ? p.v[c0 + 1] : 0.f;
    vq[s].z = c0 + 2 < p.m ? p.v[c0 + 2] : 0.f;
    vq[s].w = c0 + 3 < p.m ? p.v[c0 + 3] : 0.f;
    cb[s] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    cbi[s] = make_int4(0, 0, 0, 0);
  }
  const int n_slabs = (p.n + OT_R - 1) / OT_R;
  for (int slab = blockIdx.x; slab < n_slabs; slab += p.G) {
    const int r0 = slab * OT_R;
#pragma unroll
    for (int r = 0; r < OT_R; ++r) {
      const int row = r0 + r;
      float bv = -INFINITY;
      int bi = 0x7fffffff;
      if (row < p.n) {
        const float ui = p.u[row];
        const float* zr = p.z + (int64_t)row * p.ld;
#pragma unroll
        for (int s = 0; s < CPT; ++s) {
          const int c0 = 4 * (t + blockDim.x * s);
          const float4 z = *(const float4*)(zr + (c0 < p.m ? c0 : 0));     // rows are padded to 4 floats: in bounds
          const float zz[4] = {z.x, z.y, z.z, z.w};
          const float vv[4] = {vq[s].x, vq[s].y, vq[s].z, vq[s].w};
          float* cbp = (float*)&cb[s];
          int* cip = (int*)&cbi[s];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (c0 + k < p.m) {
              const float tv = ((zz[k] + ui) + vv[k]) - p.norm;
              if (tv > bv) { bv = tv; bi = c0 + k; }
              if (tv > cbp[k]) { cbp[k] = tv; cip[k] = row; }
            }
          }
        }
      }
      // wave argmax (ties -> lower column)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      if (lane == 0) { rv[wave][r] = bv; ri[wave][r] = bi; }
    }
    __syncthreads();
    if (t < OT_R && r0 + t < p.n) {
      float bv = rv[0][t];
      int bi = ri[0][t];
      for (int w = 1; w < nw; ++w) {
        const float ov = rv[w][t];
        const int oi = ri[w][t];
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      p.max0[r0 + t] = bv;
      p.idx0[r0 + t] = bi;
    }
    __syncthreads();
  }
  float* pv = p.cbest_val + (int64_t)blockIdx.x * p.m;
  int* pi = p.cbest_idx + (int64_t)blockIdx.x * p.m;
#pragma unroll
  for (int s = 0; s < CPT; ++s) {
    const int c0 = 4 * (t + blockDim.x * s);
    const float* cbp = (const float*)&cb[s];
    const int* cip = (const int*)&cbi[s];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (c0 + k < p.m) { pv[c0 + k] = cbp[k]; pi[c0 + k] = cip[k]; }
  }
}

__global__ __launch_bounds__(256) void ot_colbest_kernel(const OtDev* __restrict__ probs) {
  const OtDev p = probs[blockIdx.y];
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= p.m) return;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int b = 0; b < p.G; ++b) {
    const float ov = p.cbest_val[(int64_t)b * p.m + col];
    const int oi = p.cbest_idx[(int64_t)b * p.m + col];
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  p.max1[col] = bv;
  p.idx1[col] = bi;
}

// gmatcher.py:286-294
__global__ __launch_bounds__(256) void ot_mutual_kernel(const OtDev* __restrict__ probs, float thr) {
  const OtDev p = probs[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool bad = p.status[0] != 0.f;
  if (i < p.n) {
    const int j = p.idx0[i];
    const bool mutual = (unsigned)j < (unsigned)p.m && p.idx1[j] == i;
    const float sc = mutual ? expf(p.max0[i]) : 0.f;
    p.mscores0[i] = sc;
    p.matches0[i] = (!bad && mutual && sc > thr) ? (int64_t)j : (int64_t)-1;
  }
  if (i < p.m) {
    const int r = p.idx1[i];
    const bool rok = (unsigned)r < (unsigned)p.n;
    const bool mutual1 = rok && p.idx0[r] == i;
    // mscores1 = where(mutual1, mscores0[idx1], 0); valid1 = mutual1 & valid0[idx1]
    const int jr = rok ? p.idx0[r] : -1;
    const bool mutual0_r = rok && (unsigned)jr < (unsigned)p.m && p.idx1[jr] == r;
    const float sc0 = mutual0_r ? expf(p.max0[r]) : 0.f;
    p.mscores1[i] = mutual1 ? sc0 : 0.f;
    const bool valid0_r = mutual0_r && sc0 > thr;
    p.matches1[i] = (!bad && mutual1 && valid0_r) ? (int64_t)r : (int64_t)-1;
  }
}

__global__ void ot_matrix_kernel(const float* __restrict__ z, int64_t ld, int n, int m, float alpha,
                                 const float* __restrict__ uv, float* __restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j > m || i > n) return;
  const float norm = -logf((float)n + (float)m);
  const float zz = (i < n && j < m) ? z[(int64_t)i * ld + j] : alpha;
  out[(int64_t)i * (m + 1) + j] = ((zz + uv[i]) + uv[n + 1 + j]) - norm;
}

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

static void ot_launch_shape(const gims_ot_problem* pr, int np, int& threads, int& cpt, int& maxn, int& maxm) {
  maxn = 0; maxm = 0;
  for (int i = 0; i < np; ++i) { maxn = pr[i].n > maxn ? pr[i].n : maxn; maxm = pr[i].m > maxm ? pr[i].m : maxm; }
  const int quads = (maxm + 3) / 4;
  threads = ((quads + 63) / 64) * 64;
  if (threads > 1024) threads = 1024;
  if (threads < 64) threads = 64;
  cpt = (quads + threads - 1) / threads;
}
// workgroups per problem.  Large workgroups (one resident per CU) run persistent-style: ~256 in total, each
// walking several slabs with the prefetch above; small workgroups want ~4 per CU for latency hiding.
static int ot_G(int n, int np, int threads, int cpt) {
  const int rows = cpt >= 4 ? 2 : (cpt == 2 ? 4 : OT_R);   // rows per slab of ot_iter_kernel<CPT, R>
  const int total = threads >= 1024 ? 256 : (threads >= 512 ? 512 : 1024);
  int cap = total / (np > 0 ? np : 1);
  if (cap < 4) cap = 4;
  int g = (n + rows - 1) / rows;
  return g < cap ? g : cap;
}
static size_t ot_problem_bytes(const gims_ot_problem& q, int G) {
  size_t b = 0;
  b += al256((size_t)G * (q.m + 1) * 4);          // partial
  b += al256((size_t)G * q.m * 4) * 2;            // cbest val/idx
  b += al256((size_t)q.n * 4) * 2;                // max0/idx0
  b += al256((size_t)q.m * 4) * 2;                // max1/idx1
  return b;
}

}  // namespace gims

extern "C" size_t gims_sinkhorn_workspace_bytes(const gims_ot_problem* pr, int32_t np) {
  using namespace gims;
  if (!pr || np <= 0) return 0;
  int threads, cpt, maxn, maxm;
  ot_launch_shape(pr, np, threads, cpt, maxn, maxm);
  size_t b = al256(sizeof(OtDev) * (size_t)np);
  for (int i = 0; i < np; ++i) b += ot_problem_bytes(pr[i], ot_G(pr[i].n, np, threads, cpt));
  return b;
}

extern "C" int gims_sinkhorn_match(const gims_ot_problem* pr, int32_t np, float alpha, int32_t iters,
                                   float match_threshold, void* work, size_t work_bytes, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(pr && np > 0 && work, "gims_sinkhorn_match: null / empty arguments");
  GIMS_CHECK_ARG(iters >= 0, "gims_sinkhorn_match: iters < 0");
  GIMS_CHECK_ARG(work_bytes >= gims_sinkhorn_workspace_bytes(pr, np), "gims_sinkhorn_match: workspace too small (%zu < %zu)",
                 work_bytes, gims_sinkhorn_workspace_bytes(pr, np));
  hipStream_t s = (hipStream_t)stream;
  int threads, cpt, maxn, maxm;
  ot_launch_shape(pr, np, threads, cpt, maxn, maxm);
  GIMS_CHECK_ARG(cpt <= 4, "gims_sinkhorn_match: m=%d too large (max 16384)", maxm);
  std::vector<OtDev> hprob(np);
  char* base = (char*)work;
  size_t off = al256(sizeof(OtDev) * (size_t)np);
  int maxG = 0;
  for (int i = 0; i < np; ++i) {
    const gims_ot_problem& q = pr[i];
    GIMS_CHECK_ARG(q.n > 0 && q.m > 0 && q.scores && q.uv && q.matches0 && q.matches1 && q.mscores0 && q.mscores1,
                   "gims_sinkhorn_match: problem %d has empty shape or null pointer", i);
    GIMS_CHECK_ARG((q.ld % 4) == 0 && (((uintptr_t)q.scores) & 15) == 0, "gims_sinkhorn_match: scores must be 16-byte aligned with ld %% 4 == 0");
    OtDev d;
    d.z = q.scores; d.ld = q.ld; d.n = q.n; d.m = q.m;
    d.u = q.uv; d.v = q.uv + q.n + 1; d.status = q.uv + q.n + 1 + q.m + 1;
    d.G = ot_G(q.n, np, threads, cpt);
    maxG = d.G > maxG ? d.G : maxG;
    d.partial = (float*)(base + off); off += al256((size_t)d.G * (q.m + 1) * 4);
    d.cbest_val = (float*)(base + off); off += al256((size_t)d.G * q.m * 4);
    d.cbest_idx = (int*)(base + off); off += al256((size_t)d.G * q.m * 4);
    d.max0 = (float*)(base + off); off += al256((size_t)q.n * 4);
    d.idx0 = (int*)(base + off); off += al256((size_t)q.n * 4);
    d.max1 = (float*)(base + off); off += al256((size_t)q.m * 4);
    d.idx1 = (int*)(base + off); off += al256((size_t)q.m * 4);
    d.matches0 = q.matches0; d.matches1 = q.matches1; d.mscores0 = q.mscores0; d.mscores1 = q.mscores1;
    const float ms = (float)q.n, ns = (float)q.m;            // gmatcher.py:53 (m rows, n cols there)
    d.norm = -logf(ms + ns);
    d.log_mu_bin = logf(ns) + d.norm;                        // gmatcher.py:63
    d.log_nu_bin = logf(ms) + d.norm;                        // gmatcher.py:64
    hprob[i] = d;
  }
  {
    const int rc = upload_table(hprob.data(), sizeof(OtDev) * (size_t)np, work, s);   // by kernel arguments: no sync
    if (rc != GIMS_OK) return rc;
  }
  const OtDev* dp = (const OtDev*)work;
  hipLaunchKernelGGL(ot_init_kernel, dim3(cdiv(maxn, 4), np), dim3(256), 0, s, dp, alpha, iters == 0 ? 1 : 0);
  dim3 gi(maxG, np), gc(cdiv(maxm + 1, 64), np);
  for (int it = 0; it < iters; ++it) {
    if (cpt == 1) hipLaunchKernelGGL((ot_iter_kernel<1, 8>), gi, dim3(threads), 0, s, dp, alpha);
    else if (cpt == 2) hipLaunchKernelGGL((ot_iter_kernel<2, 4>), gi, dim3(threads), 0, s, dp, alpha);
    else hipLaunchKernelGGL((ot_iter_kernel<4, 2>), gi, dim3(threads), 0, s, dp, alpha);
    hipLaunchKernelGGL(ot_colreduce_kernel, gc, dim3(1024), 0, s, dp);
  }
  if (cpt == 1) hipLaunchKernelGGL(ot_select_kernel<1>, gi, dim3(threads), 0, s, dp);
  else if (cpt == 2) hipLaunchKernelGGL(ot_select_kernel<2>, gi, dim3(threads), 0, s, dp);
  else hipLaunchKernelGGL(ot_select_kernel<4>, gi, dim3(threads), 0, s, dp);
  hipLaunchKernelGGL(ot_colbest_kernel, dim3(cdiv(maxm, 256), np), dim3(256), 0, s, dp);
  const int mx = maxn > maxm ? maxn : maxm;
  hipLaunchKernelGGL(ot_mutual_kernel, dim3(cdiv(mx, 256), np), dim3(256), 0, s, dp, match_threshold);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_ot_matrix(const float* scores, int64_t ld, int32_t n, int32_t m, float alpha, const float* uv,
                              float* out, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(scores && uv && out && n > 0 && m > 0, "gims_ot_matrix: bad arguments");
  hipLaunchKernelGGL(ot_matrix_kernel, dim3(cdiv(m + 1, 256), n + 1), dim3(256), 0, (hipStream_t)stream, scores, ld, n, m,
                     alpha, uv, out);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

// Per-pair evaluation that follows the matcher in the reference's eval loop:
//   GT matching  -- torch_find_matches (utils/preprocess_utils.py:98-132): iterated mutual nearest neighbours between the
//                   warped keypoints of image 0 and the keypoints of image 1, accepted below dist_thresh;
//   precision / recall (eval_homography.py:207-209, 222-226);
//   homography through the four most confident matches (eval_homography.py:216-217, cv2.getPerspectiveTransform);
//   RANSAC homography over all matches (eval_homography.py:193, 218, cv2.findHomography) -- this build's own, fully
//   specified RANSAC (specification in include/gims_hip.h), OpenCV's is not reproducible from outside;
//   corner error against the ground-truth homography (eval_homography.py:210, 219-223; common.py:477-481).
// Everything is batched over pairs (blockIdx.y) and stays on the device: the records feed the statistics all-gather.
//
// Arithmetic that decides index sets follows the reference's float32 operation order exactly: the warp is the fma chain
// torch's CPU matmul produces (fma(h2, 1, fma(h1, y, h0 * x)), verified bit-exact against the reference's output), the
// distance is sqrt(dx*dx + dy*dy) with every operation rounded separately (no contraction), argmin keeps the first minimum.
// The homography estimation runs in float64.
#include "common.h"

#include <string.h>

#include <vector>

namespace gims {

struct EvalDev {
  const float* kp0; const float* kp1; const int64_t* matches0; const float* mscores0;
  int n0, n1, height, width;
  float hgt[9];
  int32_t* gt0; uint8_t* inlier; float* record; float* hom;
  // workspace
  float* proj; int32_t* alive0; int32_t* alive1; int32_t* min1; int32_t* min2; int32_t* gt1; int32_t* midx; int32_t* hypcount;
  int32_t* nvalid;     // [1] number of valid matches (set by the counts kernel)
};

__device__ __forceinline__ float ref_dist(float ax, float ay, float bx, float by) {
  const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by);
  return __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
}

// ---------------------------------------------------------------------------------------------- GT matching
__global__ __launch_bounds__(256) void eval_warp_kernel(const EvalDev* __restrict__ ev) {
  const EvalDev& e = ev[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < e.n0) {
    const float x = e.kp0[2 * i], y = e.kp0[2 * i + 1];
    const float X = fmaf(e.hgt[2], 1.f, fmaf(e.hgt[1], y, __fmul_rn(e.hgt[0], x)));
    const float Y = fmaf(e.hgt[5], 1.f, fmaf(e.hgt[4], y, __fmul_rn(e.hgt[3], x)));
    const float W = fmaf(e.hgt[8], 1.f, fmaf(e.hgt[7], y, __fmul_rn(e.hgt[6], x)));
    e.proj[2 * i] = __fdiv_rn(X, W);
    e.proj[2 * i + 1] = __fdiv_rn(Y, W);
    e.alive0[i] = 1;
    e.gt0[i] = -1;
    e.inlier[i] = 0;
  }
  if (i < e.n1) { e.alive1[i] = 1; e.gt1[i] = -1; }
}

// one wave per alive point of A: nearest alive point of B (first minimum).  ROWS: A = proj0, B = kp1 (min1); else A = kp1, B = proj0
template <bool ROWS>
__global__ __launch_bounds__(256) void eval_argmin_kernel(const EvalDev* __restrict__ ev) {
  const EvalDev& e = ev[blockIdx.y];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int a = blockIdx.x * 4 + wave;
  const int na = ROWS ? e.n0 : e.n1, nb = ROWS ? e.n1 : e.n0;
  if (a >= na) return;
  const int32_t* alive_a = ROWS ? e.alive0 : e.alive1;
  const int32_t* alive_b = ROWS ? e.alive1 : e.alive0;
  int32_t* out = ROWS ? e.min1 : e.min2;
  if (!alive_a[a]) { if (lane == 0) out[a] = -1; return; }
  const float* pa = ROWS ? e.proj : e.kp1;
  const float* pb = ROWS ? e.kp1 : e.proj;
  const float ax = pa[2 * a], ay = pa[2 * a + 1];
  float bd = INFINITY;
  int bj = 0x7fffffff;
  for (int j = lane; j < nb; j += 64) {
    if (!alive_b[j]) continue;
    // the reference always forms (projected keypoint of image 0) - (keypoint of image 1)
    const float d = ROWS ? ref_dist(ax, ay, pb[2 * j], pb[2 * j + 1]) : ref_dist(pb[2 * j], pb[2 * j + 1], ax, ay);
    if (d < bd) { bd = d; bj = j; }                      // ascending j: the first minimum stays
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float od = __shfl_xor(bd, o, 64);
    const int oj = __shfl_xor(bj, o, 64);
    if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
  }
  if (lane == 0) out[a] = bj == 0x7fffffff ? -1 : bj;
}

__global__ __launch_bounds__(256) void eval_mutual_kernel(const EvalDev* __restrict__ ev, float dist_thresh) {
  const EvalDev& e = ev[blockIdx.y];
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= e.n1 || !e.alive1[j]) return;
  const int i = e.min2[j];
  if (i < 0 || e.min1[i] != j) return;
  if (ref_dist(e.proj[2 * i], e.proj[2 * i + 1], e.kp1[2 * j], e.kp1[2 * j + 1]) < dist_thresh) {
    e.gt0[i] = j;
    e.gt1[j] = i;
    e.alive0[i] = 0;      // i is paired with exactly one j (mutual), so no two threads write the same slot
    e.alive1[j] = 0;
  }
}

// ---------------------------------------------------------------------------------------------- small dense algebra
// H (h[8] = 1) through 4 point pairs: Gaussian elimination with partial pivoting on the 8x8 system, float64.
__device__ bool solve8(double (&A)[8][9]) {
  for (int c = 0; c < 8; ++c) {
    int piv = c;
    double best = fabs(A[c][c]);
    for (int r = c + 1; r < 8; ++r)
      if (fabs(A[r][c]) > best) { best = fabs(A[r][c]); piv = r; }
    if (!(best > 1e-300)) return false;
    if (piv != c)
      for (int k = c; k < 9; ++k) { const double t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; }
    const double inv = 1.0 / A[c][c];
    for (int r = c + 1; r < 8; ++r) {
      const double f = A[r][c] * inv;
      if (f != 0.0)
        for (int k = c; k < 9; ++k) A[r][k] -= f * A[c][k];
    }
  }
  for (int c = 7; c >= 0; --c) {
    double s = A[c][8];
    for (int k = c + 1; k < 8; ++k) s -= A[c][k] * A[k][8];
    A[c][8] = s / A[c][c];
  }
  return true;
}

__device__ bool homography4(const float* p0, const float* p1, const int (&idx)[4], const int32_t* midx, const int64_t* matches0,
                            double (&H)[9]) {
  double A[8][9];
  for (int k = 0; k < 4; ++k) {
    const int i = midx[idx[k]];
    const int j = (int)matches0[i];
    const double x = p0[2 * i], y = p0[2 * i + 1], u = p1[2 * j], v = p1[2 * j + 1];
    const double r0[9] = {x, y, 1, 0, 0, 0, -u * x, -u * y, u};
    const double r1[9] = {0, 0, 0, x, y, 1, -v * x, -v * y, v};
    for (int c = 0; c < 9; ++c) { A[2 * k][c] = r0[c]; A[2 * k + 1][c] = r1[c]; }
  }
  if (!solve8(A)) return false;
  bool fin = true;
  for (int c = 0; c < 8; ++c) { H[c] = A[c][8]; fin = fin && isfinite(H[c]); }
  H[8] = 1.0;
  return fin;
}

__device__ __forceinline__ double reproj2(const double (&H)[9], double x, double y, double u, double v) {
  const double w = H[6] * x + H[7] * y + H[8];
  const double qx = (H[0] * x + H[1] * y + H[2]) / w, qy = (H[3] * x + H[4] * y + H[5]) / w;
  return (qx - u) * (qx - u) + (qy - v) * (qy - v);
}

// mean corner distance between two homographies (eval_homography.py:210, 219-223): corners transformed in float64,
// rounded to float32 like cv2.perspectiveTransform's output, error in float32 like compute_pixel_error
__device__ float corner_error(const double (&He)[9], const float* hgt, int height, int width) {
  const float cx[4] = {0.f, 0.f, (float)width, (float)width}, cy[4] = {0.f, (float)height, (float)height, 0.f};
  double Hg[9];
  for (int c = 0; c < 9; ++c) Hg[c] = hgt[c];
  float acc = 0.f;
  for (int k = 0; k < 4; ++k) {
    const double x = cx[k], y = cy[k];
    const double we = He[6] * x + He[7] * y + He[8], wg = Hg[6] * x + Hg[7] * y + Hg[8];
    const float ex = (float)((He[0] * x + He[1] * y + He[2]) / we), ey = (float)((He[3] * x + He[4] * y + He[5]) / we);
    const float gx = (float)((Hg[0] * x + Hg[1] * y + Hg[2]) / wg), gy = (float)((Hg[3] * x + Hg[4] * y + Hg[5]) / wg);
    const float dx = gx - ex, dy = gy - ey;
    acc += sqrtf(dx * dx + dy * dy);
  }
  return acc / 4.f;
}

__device__ __forceinline__ uint64_t splitmix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ void ransac_sample(uint64_t seed, int hyp, int k, int (&idx)[4]) {     // sampler of the specification in include/gims_hip.h
  uint64_t state = seed ^ ((uint64_t)hyp * 0xD1342543DE82EF95ull);
  int n = 0;
  while (n < 4) {
    state = splitmix(state);
    const int c = (int)(state % (uint64_t)k);
    bool dup = false;
    for (int q = 0; q < n; ++q) dup = dup || idx[q] == c;
    if (!dup) idx[n++] = c;
  }
}

// record layout (float[16])
enum { EV_NVALID = 0, EV_NGT = 1, EV_NCORRECT = 2, EV_NFN = 3, EV_PRECISION = 4, EV_RECALL = 5, EV_NINLIERS = 6, EV_ERR_DLT = 7,
       EV_ERR_RANSAC = 8, EV_DLT_OK = 9, EV_RANSAC_OK = 10 };

// ---------------------------------------------------------------------------------------------- counts, compaction, DLT
// one workgroup per pair: precision / recall counters, ascending list of valid matches, four most confident, H_dlt
__global__ __launch_bounds__(1024) void eval_counts_kernel(const EvalDev* __restrict__ ev) {
  __shared__ int s_cnt[3];
  __shared__ int s_scan[1024];
  __shared__ float s_bv[16][4];
  __shared__ int s_bi[16][4];
  const EvalDev& e = ev[blockIdx.x];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < 3) s_cnt[t] = 0;
  __syncthreads();
  int n_gt = 0, n_corr = 0, n_fn = 0;
  for (int i = t; i < e.n0; i += 1024) {
    const int64_t m = e.matches0[i];
    const int g = e.gt0[i];
    n_gt += g >= 0;
    n_corr += g >= 0 && m == (int64_t)g;
    n_fn += m == -1 && g != -1;
  }
  atomicAdd(&s_cnt[0], n_gt);
  atomicAdd(&s_cnt[1], n_corr);
  atomicAdd(&s_cnt[2], n_fn);
  // ordered compaction of the valid matches (chunks of 1024 keypoints, inclusive scan in LDS)
  int base = 0;
  for (int c0 = 0; c0 < e.n0; c0 += 1024) {
    const int i = c0 + t;
    const int v = i < e.n0 && e.matches0[i] > -1 ? 1 : 0;
    s_scan[t] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int add = t >= o ? s_scan[t - o] : 0;
      __syncthreads();
      s_scan[t] += add;
      __syncthreads();
    }
    if (v) e.midx[base + s_scan[t] - 1] = i;
    base += s_scan[1023];
    __syncthreads();
  }
  const int K = base;
  // four most confident valid matches; ties -> the earlier match
  float bv[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int bi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
  auto push = [&](float v, int p) {
    if (p == 0x7fffffff) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (v > bv[q] || (v == bv[q] && p < bi[q])) {
        const float tv = bv[q]; const int tp = bi[q];
        bv[q] = v; bi[q] = p; v = tv; p = tp;
      }
    }
  };
  __syncthreads();                                     // midx complete
  for (int p = t; p < K; p += 1024) push(e.mscores0[e.midx[p]], p);
  for (int o = 32; o > 0; o >>= 1) {                   // merge the top-4 lists across the wave
    float ov[4]; int oi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { ov[q] = __shfl_xor(bv[q], o, 64); oi[q] = __shfl_xor(bi[q], o, 64); }
#pragma unroll
    for (int q = 0; q < 4; ++q) push(ov[q], oi[q]);
  }
  if (lane == 0)
    for (int q = 0; q < 4; ++q) { s_bv[wave][q] = bv[q]; s_bi[wave][q] = bi[q]; }
  __syncthreads();
  if (t == 0) {
    for (int w = 1; w < 16; ++w)
      for (int q = 0; q < 4; ++q) push(s_bv[w][q], s_bi[w][q]);
    const float nv = (float)K, ncorr = (float)s_cnt[1], nfn = (float)s_cnt[2];
    e.nvalid[0] = K;
    e.record[EV_NVALID] = nv;
    e.record[EV_NGT] = (float)s_cnt[0];
    e.record[EV_NCORRECT] = ncorr;
    e.record[EV_NFN] = nfn;
    e.record[EV_PRECISION] = (float)((double)s_cnt[1] / (double)K);                       // 0/0 -> NaN, like NumPy
    e.record[EV_RECALL] = (float)((double)s_cnt[1] / ((double)s_cnt[1] + (double)s_cnt[2]));
    double H[9];
    bool ok = K >= 4;
    if (ok) {
      const int idx[4] = {bi[0], bi[1], bi[2], bi[3]};
      ok = homography4(e.kp0, e.kp1, idx, e.midx, e.matches0, H);
    }
    e.record[EV_DLT_OK] = ok ? 1.f : 0.f;
    e.record[EV_ERR_DLT] = ok ? corner_error(H, e.hgt, e.height, e.width) : -1.f;
    for (int c = 0; c < 9; ++c) e.hom[c] = ok ? (float)H[c] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------- RANSAC
// one wave per hypothesis: 4-point model (all lanes solve it redundantly: no divergence), inliers counted across the lanes
__global__ __launch_bounds__(256) void eval_ransac_kernel(const EvalDev* __restrict__ ev, uint64_t seed, int iters, double t2) {
  const EvalDev& e = ev[blockIdx.y];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int hyp = blockIdx.x * 4 + wave;
  const int K = e.nvalid[0];
  if (hyp >= iters) return;
  int cnt = -1;
  if (K >= 4) {
    int idx[4];
    ransac_sample(seed, hyp, K, idx);
    double H[9];
    if (homography4(e.kp0, e.kp1, idx, e.midx, e.matches0, H)) {
      cnt = 0;
      for (int p = lane; p < K; p += 64) {
        const int i = e.midx[p];
        const int j = (int)e.matches0[i];
        cnt += reproj2(H, e.kp0[2 * i], e.kp0[2 * i + 1], e.kp1[2 * j], e.kp1[2 * j + 1]) <= t2;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    }
  }
  if (lane == 0) e.hypcount[hyp] = cnt;
}

// one workgroup per pair: best hypothesis (most inliers, first such), least-squares refit on its inliers, final mask
__global__ __launch_bounds__(1024) void eval_ransac_finish_kernel(const EvalDev* __restrict__ ev, uint64_t seed, int iters, double t2) {
  __shared__ int s_best[16][2];
  __shared__ double s_H[9];
  __shared__ double s_acc[16][44];
  __shared__ int s_ok, s_cnt;
  const EvalDev& e = ev[blockIdx.x];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int K = e.nvalid[0];
  int bc = -1, bh = 0x7fffffff;
  for (int h = t; h < iters; h += 1024) {
    const int c = e.hypcount[h];
    if (c > bc) { bc = c; bh = h; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const int oc = __shfl_xor(bc, o, 64), oh = __shfl_xor(bh, o, 64);
    if (oc > bc || (oc == bc && oh < bh)) { bc = oc; bh = oh; }
  }
  if (lane == 0) { s_best[wave][0] = bc; s_best[wave][1] = bh; }
  if (t == 0) { s_ok = 0; s_cnt = 0; }
  __syncthreads();
  if (t == 0) {
    for (int w = 1; w < 16; ++w)
      if (s_best[w][0] > bc || (s_best[w][0] == bc && s_best[w][1] < bh)) { bc = s_best[w][0]; bh = s_best[w][1]; }
    if (bc >= 0 && K >= 4) {
      int idx[4];
      ransac_sample(seed, bh, K, idx);
      double H[9];
      if (homography4(e.kp0, e.kp1, idx, e.midx, e.matches0, H)) {
        for (int c = 0; c < 9; ++c) s_H[c] = H[c];
        s_ok = 1;
      }
    }
  }
  __syncthreads();
  if (!s_ok) {
    if (t == 0) {
      e.record[EV_RANSAC_OK] = 0.f; e.record[EV_NINLIERS] = 0.f; e.record[EV_ERR_RANSAC] = -1.f;
      for (int c = 0; c < 9; ++c) e.hom[9 + c] = 0.f;
    }
    return;
  }
  double H[9];
  for (int c = 0; c < 9; ++c) H[c] = s_H[c];
  // normal equations of the 2K x 8 system over the inliers of the best hypothesis: 36 entries of A^T A (upper) + 8 of A^T b
  double acc[44];
  for (int c = 0; c < 44; ++c) acc[c] = 0.0;
  int nin = 0;
  for (int p = t; p < K; p += 1024) {
    const int i = e.midx[p];
    const int j = (int)e.matches0[i];
    const double x = e.kp0[2 * i], y = e.kp0[2 * i + 1], u = e.kp1[2 * j], v = e.kp1[2 * j + 1];
    if (reproj2(H, x, y, u, v) <= t2) {
      ++nin;
      const double r0[8] = {x, y, 1, 0, 0, 0, -u * x, -u * y}, r1[8] = {0, 0, 0, x, y, 1, -v * x, -v * y};
      int q = 0;
      for (int a = 0; a < 8; ++a)
        for (int b = a; b < 8; ++b) acc[q++] += r0[a] * r0[b] + r1[a] * r1[b];
      for (int a = 0; a < 8; ++a) acc[36 + a] += r0[a] * u + r1[a] * v;
    }
  }
  for (int c = 0; c < 44; ++c) {
    double s = acc[c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) s_acc[wave][c] = s;
  }
  atomicAdd(&s_cnt, nin);
  __syncthreads();
  if (t == 0) {
    if (s_cnt >= 4) {
      double A[8][9];
      int q = 0;
      for (int a = 0; a < 8; ++a)
        for (int b = a; b < 8; ++b) {
          double s = 0.0;
          for (int w = 0; w < 16; ++w) s += s_acc[w][q];
          A[a][b] = s; A[b][a] = s;
          ++q;
        }
      for (int a = 0; a < 8; ++a) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += s_acc[w][36 + a];
        A[a][8] = s;
      }
      if (solve8(A)) {
        bool fin = true;
        for (int c = 0; c < 8; ++c) fin = fin && isfinite(A[c][8]);
        if (fin) {
          for (int c = 0; c < 8; ++c) s_H[c] = A[c][8];
          s_H[8] = 1.0;
        }
      }
    }
    s_cnt = 0;
  }
  __syncthreads();
  for (int c = 0; c < 9; ++c) H[c] = s_H[c];
  nin = 0;
  for (int p = t; p < K; p += 1024) {
    const int i = e.midx[p];
    const int j = (int)e.matches0[i];
    const bool in = reproj2(H, e.kp0[2 * i], e.kp0[2 * i + 1], e.kp1[2 * j], e.kp1[2 * j + 1]) <= t2;
    e.inlier[i] = in ? 1 : 0;
    nin += in;
  }
  atomicAdd(&s_cnt, nin);
  __syncthreads();
  if (t == 0) {
    e.record[EV_RANSAC_OK] = 1.f;
    e.record[EV_NINLIERS] = (float)s_cnt;
    e.record[EV_ERR_RANSAC] = corner_error(H, e.hgt, e.height, e.width);
    for (int c = 0; c < 9; ++c) e.hom[9 + c] = (float)H[c];
  }
}

static inline size_t al256e(size_t x) { return (x + 255) & ~(size_t)255; }
static size_t eval_pair_bytes(const gims_eval_pair& p, int iters) {
  return al256e((size_t)p.n0 * 8) + 5 * al256e((size_t)p.n0 * 4) + 3 * al256e((size_t)p.n1 * 4) + al256e((size_t)iters * 4) + 256;
}

}  // namespace gims

extern "C" size_t gims_eval_workspace_bytes(const gims_eval_pair* pairs, int32_t n_pairs, int32_t ransac_iters) {
  using namespace gims;
  if (!pairs || n_pairs <= 0 || ransac_iters < 0) return 0;
  size_t b = al256e(sizeof(EvalDev) * (size_t)n_pairs);
  for (int i = 0; i < n_pairs; ++i) b += eval_pair_bytes(pairs[i], ransac_iters);
  return b;
}

extern "C" int gims_eval_pairs(const gims_eval_pair* pairs, int32_t n_pairs, float dist_thresh, int32_t n_iters, float ransac_thresh,
                               int32_t ransac_iters, uint64_t seed, void* work, size_t work_bytes, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(pairs && n_pairs > 0 && work, "gims_eval_pairs: null / empty arguments");
  GIMS_CHECK_ARG(n_iters >= 0 && ransac_iters >= 0 && ransac_iters <= (1 << 20), "gims_eval_pairs: bad iteration counts");
  GIMS_CHECK_ARG(work_bytes >= gims_eval_workspace_bytes(pairs, n_pairs, ransac_iters), "gims_eval_pairs: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  std::vector<EvalDev> h(n_pairs);
  char* base = (char*)work;
  size_t off = al256e(sizeof(EvalDev) * (size_t)n_pairs);
  int maxn0 = 0, maxn1 = 0;
  for (int i = 0; i < n_pairs; ++i) {
    const gims_eval_pair& p = pairs[i];
    GIMS_CHECK_ARG(p.n0 > 0 && p.n1 > 0 && p.kpts0 && p.kpts1 && p.matches0 && p.mscores0 && p.gt0 && p.inlier && p.record && p.homographies,
                   "gims_eval_pairs: pair %d has an empty shape or a null pointer", i);
    EvalDev d;
    d.kp0 = p.kpts0; d.kp1 = p.kpts1; d.matches0 = p.matches0; d.mscores0 = p.mscores0;
    d.n0 = p.n0; d.n1 = p.n1; d.height = p.height; d.width = p.width;
    memcpy(d.hgt, p.h_gt, sizeof(d.hgt));
    d.gt0 = p.gt0; d.inlier = p.inlier; d.record = p.record; d.hom = p.homographies;
    d.proj = (float*)(base + off); off += al256e((size_t)p.n0 * 8);
    d.alive0 = (int32_t*)(base + off); off += al256e((size_t)p.n0 * 4);
    d.min1 = (int32_t*)(base + off); off += al256e((size_t)p.n0 * 4);
    d.midx = (int32_t*)(base + off); off += al256e((size_t)p.n0 * 4);
    off += 2 * al256e((size_t)p.n0 * 4);       // spare
    d.alive1 = (int32_t*)(base + off); off += al256e((size_t)p.n1 * 4);
    d.min2 = (int32_t*)(base + off); off += al256e((size_t)p.n1 * 4);
    d.gt1 = (int32_t*)(base + off); off += al256e((size_t)p.n1 * 4);
    d.hypcount = (int32_t*)(base + off); off += al256e((size_t)ransac_iters * 4);
    d.nvalid = (int32_t*)(base + off); off += 256;
    h[i] = d;
    maxn0 = p.n0 > maxn0 ? p.n0 : maxn0;
    maxn1 = p.n1 > maxn1 ? p.n1 : maxn1;
  }
  int rc = upload_table(h.data(), sizeof(EvalDev) * (size_t)n_pairs, work, s);
  if (rc != GIMS_OK) return rc;
  const EvalDev* dev = (const EvalDev*)work;
  const int mx = maxn0 > maxn1 ? maxn0 : maxn1;
  hipLaunchKernelGGL(eval_warp_kernel, dim3(cdiv(mx, 256), n_pairs), dim3(256), 0, s, dev);
  for (int it = 0; it < n_iters; ++it) {
    hipLaunchKernelGGL(eval_argmin_kernel<true>, dim3(cdiv(maxn0, 4), n_pairs), dim3(256), 0, s, dev);
    hipLaunchKernelGGL(eval_argmin_kernel<false>, dim3(cdiv(maxn1, 4), n_pairs), dim3(256), 0, s, dev);
    hipLaunchKernelGGL(eval_mutual_kernel, dim3(cdiv(maxn1, 256), n_pairs), dim3(256), 0, s, dev, dist_thresh);
  }
  hipLaunchKernelGGL(eval_counts_kernel, dim3(n_pairs), dim3(1024), 0, s, dev);
  const double t2 = (double)ransac_thresh * (double)ransac_thresh;
  if (ransac_iters > 0) hipLaunchKernelGGL(eval_ransac_kernel, dim3(cdiv(ransac_iters, 4), n_pairs), dim3(256), 0, s, dev, seed, ransac_iters, t2);
  hipLaunchKernelGGL(eval_ransac_finish_kernel, dim3(n_pairs), dim3(1024), 0, s, dev, seed, ransac_iters, t2);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

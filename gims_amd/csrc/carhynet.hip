// CAR-HyNet patch descriptor: the layers of /root/reference/carhynet/models.py:311-399 that are not
// plain GEMMs.  Activations are NHWC f32 in HBM ([patch][y][x][channel]: a pixel's channels are contiguous, so a 3x3
// convolution is im2col + the split-bf16 GEMM of linear.hip with output rows = pixels, and the GEMM's output IS the next
// NHWC activation).  First version of this row: correct and parity-checked; the kernels below are simple streaming kernels
// (one thread per output element or per (pixel, channel quad)), HBM-bound by design, not yet fused.
//
//   gims_ch_frn_stats     FRN's per-(patch, channel) scale  weight * rsqrt(mean over H x W of x^2 + |eps|)        models.py:67-82
//   gims_ch_pool_hw       CoordAtt's two average pools (over W and over H), optionally of the FRN output    models.py:141-143
//   gims_ch_gates         CoordAtt's shared 1x1 conv + BN + h_swish and the two 1x1 convs + sigmoid         models.py:144-151
//   gims_ch_apply         y = max((x * s + b) * a_w * a_h, tau): FRN scale, CoordAtt gates, TLU in one pass models.py:78-84,152,107
//   gims_ch_im2col3       3x3 patches (pad 1, stride 1 or 2) written as SPL32 split-bf16 GEMM operand rows
//   gims_ch_dwconv3       depthwise 3x3 + folded BatchNorm (+ReLU6 on input / output, + residual)          models.py:172-180, 207, 220-223
//   gims_ch_gate_pw_pw    SandGlass middle in one pass: CoordAtt gates applied, 1x1 C->16 (+BN), 1x1 16->C (+BN, ReLU6)  models.py:152, 208-218
//   gims_ch_input_block   FRN(3) + TLU(3) on the raw patches and the first convolution's split-bf16 operand rows, one workgroup per patch  models.py:316-317
//   gims_ch_frn_block     FRN (+ CoordAtt) + TLU of one layer, one workgroup per patch: one read, one write of the activation   models.py:57-108, 139-153
//   gims_ch_sandglass     the whole SandGlass block + outer residual, one workgroup per patch, activation resident in LDS  models.py:182-235
//   gims_ch_l2norm        x / sqrt(sum x^2 + 1e-10) per row                                                 models.py:9-21
#include "common.h"

namespace gims {

// one workgroup per (patch, 32-channel group): 256 threads = 8 pixel lanes x 32 channels
__global__ __launch_bounds__(256) void ch_frn_stats_kernel(const float* __restrict__ x, int hw, int c, const float* __restrict__ wgt, float eps,
                                                           float* __restrict__ scale) {
  __shared__ float red[8][32];
  const int p = blockIdx.x, cg = blockIdx.y, cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int ch = cg * 32 + cl;
  float s = 0.f;
  if (ch < c) {
    const float* xp = x + (int64_t)p * hw * c + ch;
    for (int i = pl; i < hw; i += 8) { const float v = xp[(int64_t)i * c]; s = fmaf(v, v, s); }
  }
  red[pl][cl] = s;
  __syncthreads();
  if (pl == 0 && ch < c) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += red[j][cl];
    scale[(int64_t)p * c + ch] = wgt[ch] * rsqrtf(t / (float)hw + eps);      // x * rsqrt(nu2 + |eps|) * weight (models.py:67-82)
  }
}

// pooled means over W (ph[p][y][c]) and over H (pw[p][x][c]) of  x * s[p][c] + b[c]  (s, b may be null: identity).
// One thread per output element: blockIdx.y = 0 -> ph (sum over x), 1 -> pw (sum over y); consecutive threads = consecutive
// channels, so every load of the 32-step loop is a contiguous line across the wave.  The patch is read twice (once per
// direction), from L2 the second time.
__global__ __launch_bounds__(256) void ch_pool_hw_kernel(const float* __restrict__ x, int64_t patches, int h, int w, int c, const float* __restrict__ s,
                                                         const float* __restrict__ b, float* __restrict__ ph, float* __restrict__ pw,
                                                         float* __restrict__ rowsq) {
  const bool over_x = blockIdx.y == 0;
  const int n_line = over_x ? h : w, n_sum = over_x ? w : h;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= patches * n_line * c) return;
  const int ch = (int)(i % c);
  const int line = (int)((i / c) % n_line);
  const int64_t p = i / ((int64_t)c * n_line);
  const float* xp = x + (p * h * w + (over_x ? (int64_t)line * w : line)) * c + ch;
  const int64_t step = over_x ? c : (int64_t)w * c;
  float t = 0.f, q = 0.f;
  for (int k = 0; k < n_sum; ++k) { const float v = xp[k * step]; t += v; q = fmaf(v, v, q); }
  t /= (float)n_sum;
  if (s) t = fmaf(t, s[p * c + ch], b ? b[ch] : 0.f);      // the mean of an affine map is the affine map of the mean
  (over_x ? ph : pw)[i] = t;
  if (rowsq && over_x) rowsq[i] = q;                        // per-row sums of squares: FRN's statistics from the same pass
}

// FRN scale from the per-row sums of squares of gims_ch_pool_hw: scale[p][c] = weight[c] * rsqrt(sum_y rowsq[p][y][c] / (h w) + eps)
__global__ __launch_bounds__(256) void ch_frn_from_rows_kernel(const float* __restrict__ rowsq, int64_t patches, int h, int w, int c, const float* __restrict__ wgt,
                                                               float eps, float* __restrict__ scale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= patches * c) return;
  const int ch = (int)(i % c);
  const int64_t p = i / c;
  float t = 0.f;
  for (int y = 0; y < h; ++y) t += rowsq[(p * h + y) * c + ch];
  scale[i] = wgt[ch] * rsqrtf(t / (float)(h * w) + eps);
}

// CoordAtt gates for one patch per workgroup: rows r = 0..h-1 from ph, h..h+w-1 from pw (optionally the pools of the RAW
// activation, mapped through FRN's per-(patch, channel) affine  v * s[p][k] + b[k]  on the way in).
//   mid[r][m] = h_swish(bn(conv1(row r)))  (8 channels; BatchNorm folded into w1 / b1 by the caller)
//   a_h[y][ch] = sigmoid(conv_h(mid[y])),  a_w[x][ch] = sigmoid(conv_w(mid[h + x]))
__global__ __launch_bounds__(256) void ch_gates_kernel(const float* __restrict__ ph, const float* __restrict__ pw, int h, int w, int c,
                                                       const float* __restrict__ w1, const float* __restrict__ b1,      // [8][c], [8]
                                                       const float* __restrict__ wh, const float* __restrict__ bh,      // [c][8], [c]
                                                       const float* __restrict__ ww, const float* __restrict__ bw,
                                                       const float* __restrict__ fs, const float* __restrict__ fb,
                                                       float* __restrict__ ah, float* __restrict__ aw) {
  __shared__ float mid[64][8];
  const int p = blockIdx.x, t = threadIdx.x, rows = h + w;
  for (int i = t; i < rows * 8; i += 256) {
    const int r = i >> 3, m = i & 7;
    const float* src = r < h ? ph + ((int64_t)p * h + r) * c : pw + ((int64_t)p * w + (r - h)) * c;
    float acc = b1[m];
    for (int k = 0; k < c; ++k) acc = fmaf(fs ? fmaf(src[k], fs[(int64_t)p * c + k], fb[k]) : src[k], w1[m * c + k], acc);
    mid[r][m] = acc * (fminf(fmaxf(acc + 3.f, 0.f), 6.f) / 6.f);
  }
  __syncthreads();
  for (int i = t; i < rows * c; i += 256) {
    const int r = i / c, ch = i - r * c;
    const bool is_h = r < h;
    const float* wt = (is_h ? wh : ww) + ch * 8;
    float acc = (is_h ? bh : bw)[ch];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc = fmaf(mid[r][m], wt[m], acc);
    const float g = 1.f / (1.f + __expf(-acc));
    if (is_h) ah[((int64_t)p * h + r) * c + ch] = g;
    else aw[((int64_t)p * w + (r - h)) * c + ch] = g;
  }
}

// four consecutive channels of one pixel as split-bf16 (SPL32: hi at o, lo at o + 32)
__device__ __forceinline__ void store_split4(uint16_t* o, const float (&v)[4]) {
  const uint32_t h01 = pack_bf2(v[0], v[1]), h23 = pack_bf2(v[2], v[3]);
  const uint32_t l01 = pack_bf2(v[0] - __uint_as_float(h01 << 16), v[1] - __uint_as_float(h01 & 0xffff0000u));
  const uint32_t l23 = pack_bf2(v[2] - __uint_as_float(h23 << 16), v[3] - __uint_as_float(h23 & 0xffff0000u));
  *(uint2*)o = make_uint2(h01, h23);
  *(uint2*)(o + 32) = make_uint2(l01, l23);
}

// y[p][y][x][ch] = max((x * s[p][ch] + b[ch]) * ah[p][y][ch] * aw[p][x][ch], tau[ch]);  any of s/b, ah/aw, tau may be null.
// Output as f32 NHWC (y) and / or as SPL32 split-bf16 pixel rows (ysp, pitch ldsp): the operand of the next convolution.
__global__ __launch_bounds__(256) void ch_apply_kernel(const float* __restrict__ x, int64_t total, int h, int w, int c, const float* __restrict__ s,
                                                       const float* __restrict__ b, const float* __restrict__ ah, const float* __restrict__ aw,
                                                       const float* __restrict__ tau, float* __restrict__ y, uint16_t* __restrict__ ysp, int64_t ldsp) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= total) return;
  const int ch = (int)(i4 % c);
  const int64_t pix = i4 / c;
  const int xx = (int)(pix % w), yy = (int)((pix / w) % h);
  const int64_t p = pix / ((int64_t)w * h);
  float4 v = *(const float4*)(x + i4);
  float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float t = r[j];
    if (s) t = fmaf(t, s[p * c + ch + j], b ? b[ch + j] : 0.f);
    if (ah) t = t * aw[(p * w + xx) * c + ch + j] * ah[(p * h + yy) * c + ch + j];
    if (tau) t = fmaxf(t, tau[ch + j]);
    r[j] = t;
  }
  if (y) *(float4*)(y + i4) = make_float4(r[0], r[1], r[2], r[3]);
  if (ysp) store_split4(ysp + pix * ldsp + spl_col(ch), r);
}

// 3x3 neighbourhoods (pad 1) of an NHWC activation as rows of a split-bf16 (SPL32) GEMM operand:
// out row = output pixel (p, yo, xo), logical column k = (ky * 3 + kx) * c + ch, K = 9 c zero-padded to kpad (multiple of 32).
__global__ __launch_bounds__(256) void ch_im2col3_kernel(const float* __restrict__ x, int64_t rows, int h, int w, int c, int stride, int ho, int wo,
                                                         int kpad, uint16_t* __restrict__ out, int64_t ld) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int kq = kpad / 4;                                   // column quads per row
  if (i >= rows * kq) return;
  const int64_t row = i / kq;
  const int k = (int)(i - row * kq) * 4;
  const int xo = (int)(row % wo), yo = (int)((row / wo) % ho);
  const int64_t p = row / ((int64_t)wo * ho);
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int kk = k + j;
    float t = 0.f;
    if (kk < 9 * c) {
      const int tap = kk / c, ch = kk - tap * c;
      const int yy = yo * stride + tap / 3 - 1, xx = xo * stride + tap % 3 - 1;
      if (yy >= 0 && yy < h && xx >= 0 && xx < w) t = x[((p * h + yy) * w + xx) * c + ch];
    }
    v[j] = t;
  }
  const uint32_t h01 = pack_bf2(v[0], v[1]), h23 = pack_bf2(v[2], v[3]);
  const uint32_t l01 = pack_bf2(v[0] - __uint_as_float(h01 << 16), v[1] - __uint_as_float(h01 & 0xffff0000u));
  const uint32_t l23 = pack_bf2(v[2] - __uint_as_float(h23 << 16), v[3] - __uint_as_float(h23 & 0xffff0000u));
  uint16_t* o = out + row * ld + spl_col(k);
  *(uint2*)o = make_uint2(h01, h23);
  *(uint2*)(o + 32) = make_uint2(l01, l23);
}

// depthwise 3x3 (pad 1, stride 1) with BatchNorm folded into wt [9][c] / bias [c]; optional ReLU6 on the output;
// optional residual: y = res_scale * res + conv.
__global__ __launch_bounds__(256) void ch_dwconv3_kernel(const float* __restrict__ x, int64_t total, int h, int w, int c, const float* __restrict__ wt,
                                                         const float* __restrict__ bias, int relu6_out, const float* __restrict__ res, float res_scale,
                                                         float* __restrict__ y, uint16_t* __restrict__ ysp, int64_t ldsp) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= total) return;
  const int ch = (int)(i4 % c);
  const int64_t pix = i4 / c;
  const int xx = (int)(pix % w), yy = (int)((pix / w) % h);
  const int64_t p = pix / ((int64_t)w * h);
  float acc[4] = {bias[ch], bias[ch + 1], bias[ch + 2], bias[ch + 3]};
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int y2 = yy + ky - 1, x2 = xx + kx - 1;
      if (y2 < 0 || y2 >= h || x2 < 0 || x2 >= w) continue;
      const float4 v = *(const float4*)(x + ((p * h + y2) * w + x2) * c + ch);
      const float4 k4 = *(const float4*)(wt + (ky * 3 + kx) * c + ch);
      acc[0] = fmaf(v.x, k4.x, acc[0]); acc[1] = fmaf(v.y, k4.y, acc[1]); acc[2] = fmaf(v.z, k4.z, acc[2]); acc[3] = fmaf(v.w, k4.w, acc[3]);
    }
  if (relu6_out) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = fminf(fmaxf(acc[j], 0.f), 6.f);
  }
  if (res) {
    const float4 r = *(const float4*)(res + i4);
    acc[0] = fmaf(r.x, res_scale, acc[0]); acc[1] = fmaf(r.y, res_scale, acc[1]); acc[2] = fmaf(r.z, res_scale, acc[2]); acc[3] = fmaf(r.w, res_scale, acc[3]);
  }
  if (y) *(float4*)(y + i4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  if (ysp) store_split4(ysp + pix * ldsp + spl_col(ch), acc);
}

// SandGlass middle, one thread per pixel: z = ReLU6(W1 (W0 (x * a_w * a_h) + b0) + b1) with C -> 16 -> C channels (C = 32 or 64),
// BatchNorm folded into W0/b0 and W1/b1 (models.py:208-218: CoordAtt output, pw-linear + BN, pw + BN + ReLU6).  The weights
// (<= 2 x 64 x 16 floats) sit in LDS; x, a_h, a_w are read as float4.
template <int C>
__global__ __launch_bounds__(256) void ch_gate_pw_pw_kernel(const float* __restrict__ x, int64_t pixels, int h, int w, const float* __restrict__ ah,
                                                            const float* __restrict__ aw, const float* __restrict__ w0, const float* __restrict__ b0,
                                                            const float* __restrict__ w1, const float* __restrict__ b1, float* __restrict__ z) {
  __shared__ float s0[16 * C], s1[C * 16], sb0[16], sb1[C];
  for (int i = threadIdx.x; i < 16 * C; i += 256) { s0[i] = w0[i]; s1[i] = w1[i]; }
  if (threadIdx.x < 16) sb0[threadIdx.x] = b0[threadIdx.x];
  if (threadIdx.x < C) sb1[threadIdx.x] = b1[threadIdx.x];
  __syncthreads();
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= pixels) return;
  const int xx = (int)(pix % w), yy = (int)((pix / w) % h);
  const int64_t p = pix / ((int64_t)w * h);
  const float* xr = x + pix * C;
  const float* hr = ah + (p * h + yy) * C;
  const float* wr = aw + (p * w + xx) * C;
  float hid[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) hid[m] = sb0[m];
#pragma unroll
  for (int k = 0; k < C; k += 4) {
    const float4 v = *(const float4*)(xr + k), g1 = *(const float4*)(hr + k), g2 = *(const float4*)(wr + k);
    const float t[4] = {v.x * g2.x * g1.x, v.y * g2.y * g1.y, v.z * g2.z * g1.z, v.w * g2.w * g1.w};      // x * a_w * a_h (models.py:152)
#pragma unroll
    for (int m = 0; m < 16; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) hid[m] = fmaf(t[j], s0[m * C + k + j], hid[m]);
  }
#pragma unroll
  for (int o = 0; o < C; o += 4) {
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float acc = sb1[o + j];
#pragma unroll
      for (int m = 0; m < 16; ++m) acc = fmaf(hid[m], s1[(o + j) * 16 + m], acc);
      r[j] = fminf(fmaxf(acc, 0.f), 6.f);
    }
    *(float4*)(z + pix * C + o) = make_float4(r[0], r[1], r[2], r[3]);
  }
}

// ---------------------------------------------------------------------------------------------- input block
// One workgroup per patch: FRN(3) + TLU(3) on the 32x32x3 input (models.py:316-317) and the 3x3 neighbourhoods of the result
// written straight as the split-bf16 operand rows of the first convolution (K = 9 taps x 4 channels -- the 4th is zero --
// padded to 64): 16 lanes write one 256-byte row, so a wave-wide store covers 1 KiB of consecutive bytes.
__global__ __launch_bounds__(256) void ch_input_block_kernel(const float* __restrict__ patches, const float* __restrict__ fw, const float* __restrict__ fb,
                                                             float eps, const float* __restrict__ tau, uint16_t* __restrict__ out, int64_t ldo) {
  __shared__ float yb[1024 * 4];
  __shared__ float red[64][4];
  __shared__ float sc[4];
  const int t = threadIdx.x;
  const float* pp = patches + (int64_t)blockIdx.x * 1024 * 3;
  float q[3] = {0.f, 0.f, 0.f};
  for (int pix = t; pix < 1024; pix += 256) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { const float v = pp[pix * 3 + c]; yb[pix * 4 + c] = v; q[c] = fmaf(v, v, q[c]); }
    yb[pix * 4 + 3] = 0.f;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q[c] += __shfl_xor(q[c], o, 64);
  }
  if ((t & 63) == 0) { red[t >> 6][0] = q[0]; red[t >> 6][1] = q[1]; red[t >> 6][2] = q[2]; }
  __syncthreads();
  if (t < 3) sc[t] = fw[t] * rsqrtf((red[0][t] + red[1][t] + red[2][t] + red[3][t]) / 1024.f + eps);
  __syncthreads();
  for (int i = t; i < 1024 * 3; i += 256) {
    const int pix = i / 3, c = i - 3 * pix;
    yb[pix * 4 + c] = fmaxf(fmaf(yb[pix * 4 + c], sc[c], fb[c]), tau[c]);
  }
  __syncthreads();
  // rows: pixel = 16 * it + (t >> 4); lane part q16 = t & 15 writes elements [8 q16, 8 q16 + 8) of the 128-element SPL32 row
  const int q16 = t & 15, blk = q16 >> 3, pos = (8 * q16) & 63, lo = pos >= 32, k0 = 32 * blk + (pos & 31);
  for (int it = 0; it < 64; ++it) {
    const int pix = 16 * it + (t >> 4), yy = pix >> 5, xx = pix & 31;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k0 + e, tap = k >> 2, c = k & 3;
      const int y2 = yy + tap / 3 - 1, x2 = xx + tap % 3 - 1;
      v[e] = (tap < 9 && y2 >= 0 && y2 < 32 && x2 >= 0 && x2 < 32) ? yb[(y2 * 32 + x2) * 4 + c] : 0.f;
    }
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t h = pack_bf2(v[2 * e], v[2 * e + 1]);
      w[e] = lo ? pack_bf2(v[2 * e] - __uint_as_float(h << 16), v[2 * e + 1] - __uint_as_float(h & 0xffff0000u)) : h;
    }
    *(uint4*)(out + ((int64_t)blockIdx.x * 1024 + pix) * ldo + 8 * q16) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// ---------------------------------------------------------------------------------------------- fused FRN (+CoordAtt) + TLU block
// One workgroup per patch: the raw convolution output [HW*HW][C] is read ONCE into LDS, FRN's statistic, CoordAtt's pools and
// gates are computed from the LDS copy, and  y = max((x s + b) a_w a_h, tau)  leaves as f32 and / or split-bf16 pixel rows:
// 8 bytes of HBM traffic per element instead of 12 (plain FRN layers) or 20 (FRN + CoordAtt layers: statistics pass, two
// pooling sweeps, apply pass).  models.py:57-85 (FRN), 139-153 (CoordAtt), 107-108 (TLU).
// position of channel c of pixel p inside the LDS image [NPIX][C] of frn_block_body
__device__ __forceinline__ int frn_slot(int p, int c, int C) { return p * C + ((((c >> 2) ^ (p & 7)) << 2) | (c & 3)); }

struct ChGateW { const float* w1; const float* b1; const float* wh; const float* bh; const float* ww; const float* bw; };   // null w1: no CoordAtt

// body shared by ch_frn_block_kernel and ch_conv_block_kernel: the raw convolution output of ONE patch is in LDS (xb [NPIX][C],
// followed by the scratch arrays); pbase = index of the patch's first pixel row in the outputs
template <int C, int HW, int NT>
__device__ __forceinline__ void frn_block_body(float* lds, const float* __restrict__ fw, const float* __restrict__ fb, float eps, const ChGateW& g,
                                               const float* __restrict__ tau, float* __restrict__ y, uint16_t* __restrict__ ysp, int64_t ldsp, int64_t pbase,
                                               unsigned long long* prof = nullptr, const float* gl = nullptr) {
  // gl: the CoordAtt gate weights staged in LDS by the caller ([w1 8C][b1 8][wh 8C][ww 8C][bh C][bw C], ch_conv_block_kernel), or null:
  // read from global memory (two L2 round trips in the middle of the block)
  constexpr int NPIX = HW * HW, QPP = C / 4, NQ = NPIX * QPP / NT, GRP = NT / C;
  auto stamp = [&](int k) __attribute__((always_inline)) { if (prof && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) prof[k] = __builtin_readcyclecounter(); };
  // xb is SWIZZLED: channel quad q of pixel p sits at quad position q ^ (p & 7) (frn_slot) -- the accumulator dump of the
  // convolution writes one pixel per lane, 128 / 256 / 512 bytes apart: unswizzled, all lanes of a store hit the same four banks
  float* xb = lds;                          // [NPIX][C]
  float* red = xb + NPIX * C;               // [GRP][C]
  float* sc = red + GRP * C;                // [C]   FRN scale of this patch
  float* ph = sc + C;                       // [HW][C] -> a_h
  float* pw = ph + HW * C;                  // [HW][C] -> a_w
  float* mid = pw + HW * C;                 // [2 HW][8]
  const int t = threadIdx.x;
  {   // FRN statistic: mean of x^2 over the pixels, per channel
    const int c = t % C, gq = t / C;
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    static_assert((NPIX / GRP) % 4 == 0, "four partial sums per thread");
#pragma unroll 2
    for (int pix = gq; pix < NPIX; pix += 4 * GRP) {          // four independent LDS reads in flight (one chain of 64 reads ran at LDS latency)
#pragma unroll
      for (int u = 0; u < 4; ++u) { const float v = xb[frn_slot(pix + u * GRP, c, C)]; s4[u] = fmaf(v, v, s4[u]); }
    }
    red[gq * C + c] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  }
  __syncthreads();
  if (t < C) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < GRP; ++q) s += red[q * C + t];
    sc[t] = fw[t] * rsqrtf(s / (float)NPIX + eps);
  }
  __syncthreads();
  stamp(4);
  const bool coord = g.w1 != nullptr;
  const float* g_w1 = gl ? gl : g.w1, *g_b1 = gl ? gl + 8 * C : g.b1, *g_wh = gl ? gl + 8 * C + 8 : g.wh, *g_ww = gl ? gl + 16 * C + 8 : g.ww,
              *g_bh = gl ? gl + 24 * C + 8 : g.bh, *g_bw = gl ? gl + 25 * C + 8 : g.bw;
  if (coord) {
    // pools of the FRN output = FRN affine map of the raw pools; one (line, channel quad) per thread: HW float4 reads
    for (int i = t; i < 2 * HW * QPP; i += NT) {
      const bool over_x = i < HW * QPP;
      const int j = over_x ? i : i - HW * QPP, line = j / QPP, q = j % QPP;
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#pragma unroll
      for (int k = 0; k < HW; k += 2) {
        const int p0 = over_x ? line * HW + k : k * HW + line, p1 = over_x ? p0 + 1 : p0 + HW;
        const float4 v0 = *(const float4*)(xb + p0 * C + ((q ^ (p0 & 7)) << 2)), v1 = *(const float4*)(xb + p1 * C + ((q ^ (p1 & 7)) << 2));
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      }
      const float4 s4 = *(const float4*)(sc + 4 * q), b4 = *(const float4*)(fb + 4 * q);
      *(float4*)((over_x ? ph : pw) + line * C + 4 * q) =
          make_float4(fmaf((a0.x + a1.x) / (float)HW, s4.x, b4.x), fmaf((a0.y + a1.y) / (float)HW, s4.y, b4.y),
                      fmaf((a0.z + a1.z) / (float)HW, s4.z, b4.z), fmaf((a0.w + a1.w) / (float)HW, s4.w, b4.w));
    }
    __syncthreads();
    for (int i = t; i < 2 * HW * 8; i += NT) {
      const int r = i >> 3, m = i & 7;
      const float* src = r < HW ? ph + r * C : pw + (r - HW) * C;
      float a4[4] = {g_b1[m], 0.f, 0.f, 0.f};
#pragma unroll 8
      for (int k = 0; k < C; k += 4) {
        const float4 wv = *(const float4*)(g_w1 + m * C + k), xv = *(const float4*)(src + k);
        a4[0] = fmaf(xv.x, wv.x, a4[0]); a4[1] = fmaf(xv.y, wv.y, a4[1]); a4[2] = fmaf(xv.z, wv.z, a4[2]); a4[3] = fmaf(xv.w, wv.w, a4[3]);
      }
      const float acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
      mid[i] = acc * (fminf(fmaxf(acc + 3.f, 0.f), 6.f) / 6.f);
    }
    __syncthreads();
    for (int i = t; i < 2 * HW * C; i += NT) {
      const bool is_h = i < HW * C;
      const int j = is_h ? i : i - HW * C, r = j / C, ch = j % C;
      const float* wt = (is_h ? g_wh : g_ww) + ch * 8;
      const float* mr = mid + (is_h ? r : HW + r) * 8;
      const float4 w0 = *(const float4*)wt, w1v = *(const float4*)(wt + 4), m0 = *(const float4*)mr, m1 = *(const float4*)(mr + 4);
      float acc = (is_h ? g_bh : g_bw)[ch];
      acc = fmaf(m0.x, w0.x, acc); acc = fmaf(m0.y, w0.y, acc); acc = fmaf(m0.z, w0.z, acc); acc = fmaf(m0.w, w0.w, acc);      // same order as the scalar loop
      acc = fmaf(m1.x, w1v.x, acc); acc = fmaf(m1.y, w1v.y, acc); acc = fmaf(m1.z, w1v.z, acc); acc = fmaf(m1.w, w1v.w, acc);
      (is_h ? ph : pw)[j] = 1.f / (1.f + __expf(-acc));
    }
    __syncthreads();
  }
  stamp(5);
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int i = t + NT * j, pix = i / QPP, ch = 4 * (i % QPP), yy = pix / HW, xx = pix % HW;
    const float4 v = *(const float4*)(xb + frn_slot(pix, ch, C));
    const float4 s4 = *(const float4*)(sc + ch), b4 = *(const float4*)(fb + ch);
    float r[4] = {fmaf(v.x, s4.x, b4.x), fmaf(v.y, s4.y, b4.y), fmaf(v.z, s4.z, b4.z), fmaf(v.w, s4.w, b4.w)};
    if (coord) {
      const float4 g1 = *(const float4*)(ph + yy * C + ch), g2 = *(const float4*)(pw + xx * C + ch);
      r[0] = r[0] * g2.x * g1.x; r[1] = r[1] * g2.y * g1.y; r[2] = r[2] * g2.z * g1.z; r[3] = r[3] * g2.w * g1.w;
    }
    const float4 t4 = *(const float4*)(tau + ch);
    r[0] = fmaxf(r[0], t4.x); r[1] = fmaxf(r[1], t4.y); r[2] = fmaxf(r[2], t4.z); r[3] = fmaxf(r[3], t4.w);
    if (y) *(float4*)(y + (pbase * C) + (int64_t)i * 4) = make_float4(r[0], r[1], r[2], r[3]);
    if (ysp) store_split4(ysp + (pbase + pix) * ldsp + spl_col(ch), r);
  }
  stamp(6);
}

template <int C, int HW, int NT>
__global__ __launch_bounds__(NT) void ch_frn_block_kernel(const float* __restrict__ x, const float* __restrict__ fw, const float* __restrict__ fb, float eps,
                                                           ChGateW g, const float* __restrict__ tau, float* __restrict__ y, uint16_t* __restrict__ ysp,
                                                           int64_t ldsp) {
  constexpr int NPIX = HW * HW, QPP = C / 4, NQ = NPIX * QPP / NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int t = threadIdx.x;
  const int64_t pbase = (int64_t)blockIdx.x * NPIX;
  const float* xp = x + pbase * C;
  {   // bulk load: NQ independent, fully coalesced 16-byte loads per thread
    float4 v[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) v[j] = *(const float4*)(xp + (int64_t)(t + NT * j) * 4);
#pragma unroll
    for (int j = 0; j < NQ; ++j) { const int i = t + NT * j; *(float4*)(lds + frn_slot(i / QPP, 4 * (i % QPP), C)) = v[j]; }
  }
  __syncthreads();
  frn_block_body<C, HW, NT>(lds, fw, fb, eps, g, tau, y, ysp, ldsp, pbase);
}

// ---------------------------------------------------------------------------------------------- fused 3x3 convolution + FRN (+CoordAtt) + TLU
// One workgroup (8 waves) per patch.  The split-bf16 input activation of the patch ([HIN*HIN][2 CIN] SPL32 pixel rows) is
// read ONCE into LDS with a one-pixel zero border, and the 3x3 convolution runs there as an implicit GEMM on the matrix
// cores: rows = output pixels, K = 9 taps x CIN channels, three bf16 MFMAs per product (hi*hi + hi*lo + lo*hi, like the
// GEMMs it replaces).  A tap is a shifted view of the LDS image -- where the gather-mode GEMM (GIMS_LINEAR_CONV3) DMA-ed every
// pixel row nine times from L2.  The weights come pre-packed in MFMA fragment order (one contiguous KiB per fragment),
// straight from L1 / L2 into registers, one K step ahead.  The accumulators (+ bias) then go to LDS over the dead input image
// and the FRN (+ CoordAtt) + TLU block runs on them as in ch_frn_block_kernel: the raw convolution output never reaches HBM.
// 16-byte chunks of a pixel record are XOR-swizzled by the pixel index so that the fragment reads of 16 consecutive pixels
// hit distinct banks.
template <int CIN, int COUT, int HIN, int STRIDE, int PP = 1>
struct ConvGeom {
  // PP patches per workgroup: the deep layers stream their whole weight set (295 / 590 KB) from L2 per patch -- 8192 patches of the 8x8x128
  // layer in 537 us are 9 TB/s of L2 -> L1 traffic, the bound of that launch; with PP patches side by side in LDS a wave owns PP times as
  // many pixel blocks of its channel block and every weight fragment serves PP times as many MFMAs.
  static constexpr int NT = 512, WAVES = 8;
  static constexpr int HOUT = (HIN - 1) / STRIDE + 1, NPIX = HOUT * HOUT, ZQ = PP * HIN * HIN;   // ZQ: index of the all-zero pixel record
  static constexpr int PXB = CIN * 4, CPP = PXB / 16;                 // bytes / 16-byte chunks per pixel record (hi + lo planes)
  static constexpr int MB = PP * NPIX / 32, NB = COUT / 32, KS = CIN / 16;
  // wave tiling: MBW m-blocks x ONE n-block per wave (the weight fragments come from global memory through L1: a wave that
  // owns several pixel blocks of one channel block loads each weight fragment once for all of them; the pixel fragments
  // come from LDS, where re-reads are cheap)
  static constexpr int NBW = 1;
  static constexpr int MBW = MB * NB / WAVES;                          // L2: 4, L3/L4: 2, L5/L6: 1
  static constexpr int IN_BYTES = (PP * HIN * HIN + 1) * PXB;
  static constexpr int FRN_FLOATS = NPIX * COUT + (NT / COUT) * COUT + COUT + 2 * HOUT * COUT + 16 * HOUT;     // per patch
  static constexpr int LDS_BYTES = IN_BYTES > PP * FRN_FLOATS * 4 ? IN_BYTES : PP * FRN_FLOATS * 4;
  static_assert(NPIX % 32 == 0, "a 32-pixel block belongs to one patch");
  static constexpr int GATE_BYTES = (26 * COUT + 8) * 4;               // staged CoordAtt gate weights (layers with CoordAtt only), behind FRN_FLOATS
  static_assert(MB * NB == WAVES * MBW * NBW, "blocks divide over the waves");
  static constexpr int HB = (CIN < 32 ? CIN : 32) / 8;                 // 16-byte chunks per plane of a channel block (32-channel blocks; 16 for the first layer)
  __device__ static __forceinline__ int swz(int q) { return CIN == 16 ? ((q >> 2) & 3) : (CIN == 32 ? ((q >> 1) & 7) : (q & 15)); }
  // hi chunk of channels [16 ks + 8 lh, +8) inside the pixel record; the lo chunk is HB further
  __device__ static __forceinline__ int chunk(int ks, int lh) { return (ks / (HB / 2)) * (2 * HB) + 2 * (ks % (HB / 2)) + lh; }
};
// Start-time stagger of the per-patch kernels (GIMS_CH_STAGGER = cycles per slot; 0 = off).  Every workgroup of these kernels takes the
// same time and all 256 CUs start together, so their HBM phases (patch load, result store) coincide: 256 x 128 KB arrive as one burst at
// the HBM rate while the memory idles during the compute phases.  Delaying workgroup b of the FIRST resident wave by ((b / 8) % 8) slots
// spreads the bursts; later workgroups inherit the offsets from the ones they replace.
__device__ __forceinline__ void ch_stagger(int cycles, int first_wave) {
  if (cycles > 0 && (int)blockIdx.x < first_wave) {
    const long long wait = (long long)(((int)blockIdx.x >> 3) & 7) * cycles;
    const long long t0 = (long long)__builtin_readcyclecounter();
    while ((long long)__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(16);
  }
}
struct ChFirst { const float* fw0; const float* fb0; const float* tau0; float eps0; unsigned long long* prof; int stagger, first_wave; };   // prof: GIMS_CH_PROF=1 phase stamps, else null   // FRN(3) + TLU(3) in front of the first convolution

// FIRST (layer 1, models.py:316-323): xin is the raw f32 patch [32*32][3]; FRN(3) + TLU(3) run here and the result is written
// into the LDS image as 16-channel split-bf16 pixel records (channels 3-15 zero) -- no operand rows through HBM at all.
template <int CIN, int COUT, int HIN, int STRIDE, bool FIRST = false, int PP = 1>
__global__ __launch_bounds__(512) void ch_conv_block_kernel(const uint16_t* __restrict__ xin, int64_t ldx, const uint16_t* __restrict__ wpk,
                                                            const float* __restrict__ bias, const float* __restrict__ fw, const float* __restrict__ fb, float eps,
                                                            ChGateW g, const float* __restrict__ tau, float* __restrict__ y, uint16_t* __restrict__ ysp,
                                                            int64_t ldsp, ChFirst first, int n_patches) {
  using G = ConvGeom<CIN, COUT, HIN, STRIDE, PP>;
  static_assert(!FIRST || PP == 1, "the first layer takes one patch per workgroup");
  static_assert(!FIRST || (CIN == 16 && HIN == 32 && STRIDE == 1), "the first layer is 32x32x3 -> 16-channel records");
  constexpr int NT = G::NT, ZQ = G::ZQ, PXB = G::PXB, CPP = G::CPP, HOUT = G::HOUT, MBW = G::MBW, NBW = G::NBW, KS = G::KS, NB = G::NB;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* img = (char*)lds;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int64_t patch = (int64_t)blockIdx.x * PP;                 // first patch of this workgroup
  const int valid = n_patches - (int)patch < PP ? n_patches - (int)patch : PP;      // patches that exist (the last workgroup of a launch)
  auto stamp = [&](int k) __attribute__((always_inline)) { if (first.prof && blockIdx.x == gridDim.x / 2 && t == 0) first.prof[k] = __builtin_readcyclecounter(); };
  ch_stagger(first.stagger, first.first_wave);
  stamp(0);
  // (a persistent patch loop, as in ch_sandglass_kernel, was measured here and dropped: the loop's live ranges took the 32x32 layers to 256
  // registers with spills and the smaller layers out of their two / three workgroups per CU -- 841 -> 878 us per 8192 patches of layer 2)
  // CoordAtt gate weights: requested before the patch, written to LDS (behind the FRN block's arrays) after it -- their latency hides
  // under the patch load instead of costing two L2 round trips inside the FRN block
  const bool stage_gates = g.w1 != nullptr && 8 * COUT <= NT;
  const bool stage_now = stage_gates;
  float gv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (stage_now) {
    if (t < 8 * COUT) { gv[0] = g.w1[t]; gv[1] = g.wh[t]; gv[2] = g.ww[t]; }
    if (t < COUT) { gv[3] = g.bh[t]; gv[4] = g.bw[t]; }
    if (t < 8) gv[5] = g.b1[t];
  }

  // ---- input patch -> LDS (swizzled chunks) + ONE all-zero pixel record that every out-of-image tap reads (no border in LDS:
  // the 16x16x64 and 8x8x128 layers then fit two / three workgroups per CU)
  if (FIRST) {
    float* yb = lds + (G::IN_BYTES + 255) / 256 * 64;            // raw patch [1024][4] behind the image (the LDS block is sized for the FRN stage)
    float* red = yb + 4096;                                       // [8 waves][4]
    float* sc0 = red + 32;
    if (t < CPP) *(uint4*)(img + ZQ * PXB + t * 16) = make_uint4(0u, 0u, 0u, 0u);
    const float* pp = (const float*)xin + patch * 3072;
    float q3[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int pix = t + NT * u;
#pragma unroll
      for (int c = 0; c < 3; ++c) { const float v = pp[pix * 3 + c]; yb[pix * 4 + c] = v; q3[c] = fmaf(v, v, q3[c]); }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) q3[c] = wave_sum(q3[c]);
    if (lane == 0) { red[wave * 4] = q3[0]; red[wave * 4 + 1] = q3[1]; red[wave * 4 + 2] = q3[2]; }
    __syncthreads();
    if (t < 3) {
      float sq = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) sq += red[w8 * 4 + t];
      sc0[t] = first.fw0[t] * rsqrtf(sq / 1024.f + first.eps0);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = t + NT * u, sw = G::swz(q);
      float v[4];
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = fmaxf(fmaf(yb[q * 4 + c], sc0[c], first.fb0[c]), first.tau0[c]);
      v[3] = 0.f;
      const uint32_t h01 = pack_bf2(v[0], v[1]), h23 = pack_bf2(v[2], 0.f);
      const uint32_t l01 = pack_bf2(v[0] - __uint_as_float(h01 << 16), v[1] - __uint_as_float(h01 & 0xffff0000u));
      const uint32_t l23 = pack_bf2(v[2] - __uint_as_float(h23 << 16), 0.f);
      const uint4 z = make_uint4(0u, 0u, 0u, 0u);
      *(uint4*)(img + q * PXB + ((0 ^ sw) * 16)) = make_uint4(h01, h23, 0u, 0u);      // hi, channels 0-7
      *(uint4*)(img + q * PXB + ((1 ^ sw) * 16)) = z;                                  // hi, channels 8-15
      *(uint4*)(img + q * PXB + ((2 ^ sw) * 16)) = make_uint4(l01, l23, 0u, 0u);      // lo, channels 0-7
      *(uint4*)(img + q * PXB + ((3 ^ sw) * 16)) = z;
    }
  } else {
    if (t < CPP) *(uint4*)(img + ZQ * PXB + t * 16) = make_uint4(0u, 0u, 0u, 0u);
    const uint16_t* src = xin + patch * (HIN * HIN) * ldx;
    constexpr int TOT = PP * HIN * HIN * CPP;
    const int last_row = valid * HIN * HIN - 1;                    // rows of missing patches re-read the last existing row (results discarded)
    constexpr int LPT = TOT / NT >= 16 ? 16 : (TOT / NT >= 8 ? 8 : 4);   // independent 16-byte loads in flight per thread: the whole patch in ONE round trip
    static_assert(TOT % (LPT * NT) == 0, "whole trips of LPT loads per thread");
#pragma unroll
    for (int i0 = t; i0 < TOT; i0 += LPT * NT) {                  // (four in flight left 4 dependent HBM round trips on the 128 KB layers)
      uint4 v[LPT];
#pragma unroll
      for (int u = 0; u < LPT; ++u) {
        const int i = i0 + u * NT;
        const int row = PP == 1 ? i / CPP : (i / CPP < last_row ? i / CPP : last_row);
        v[u] = *(const uint4*)(src + (int64_t)row * ldx + (i % CPP) * 8);
      }
#pragma unroll
      for (int u = 0; u < LPT; ++u) {
        const int i = i0 + u * NT;
        const int q = i / CPP, ch = i % CPP;
        *(uint4*)(img + q * PXB + ((ch ^ G::swz(q)) * 16)) = v[u];
      }
    }
  }
  float* gl = lds + PP * G::FRN_FLOATS;
  if (stage_now) {
    if (t < 8 * COUT) { gl[t] = gv[0]; gl[8 * COUT + 8 + t] = gv[1]; gl[16 * COUT + 8 + t] = gv[2]; }
    if (t < COUT) { gl[24 * COUT + 8 + t] = gv[3]; gl[25 * COUT + 8 + t] = gv[4]; }
    if (t < 8) gl[8 * COUT + t] = gv[5];
  }
  __syncthreads();
  stamp(1);

  // ---- implicit GEMM: this wave's (m-block, n-block) tiles
  const int nb0 = wave % NB, mb0 = (wave / NB) * MBW;            // waves with the same n-block walk consecutive pixel blocks
  int y0[MBW], x0[MBW], qb[MBW];                                 // input coordinates of tap (0, 0) of this lane's output pixels (may be -1), first record of their patch
#pragma unroll
  for (int m = 0; m < MBW; ++m) {
    const int pg = (mb0 + m) * 32 + li, p = pg % G::NPIX;
    y0[m] = (p / HOUT) * STRIDE - 1;
    x0[m] = (p % HOUT) * STRIDE - 1;
    qb[m] = (pg / G::NPIX) * (HIN * HIN);
  }
  f32x16 acc[MBW][NBW];
#pragma unroll
  for (int m = 0; m < MBW; ++m)
#pragma unroll
    for (int n = 0; n < NBW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  // weight fragments: [step = tap * KS + ks][nb][plane][lane][8]
  const uint16_t* wl = wpk + lane * 8;
  auto wfrag = [&](int step, int nb, int plane) __attribute__((always_inline)) {
    return *(const bf16x8*)(wl + (((int64_t)step * NB + nb) * 2 + plane) * 512);
  };
  constexpr int STEPS = 9 * KS;                                  // even (KS = 2, 4, 8)
  auto load_w = [&](int step, bf16x8 (&h)[NBW], bf16x8 (&l)[NBW]) __attribute__((always_inline)) {
#pragma unroll
    for (int n = 0; n < NBW; ++n) { h[n] = wfrag(step, nb0 + n, 0); l[n] = wfrag(step, nb0 + n, 1); }
  };
  auto compute = [&](int step, const bf16x8 (&h)[NBW], const bf16x8 (&l)[NBW]) __attribute__((always_inline)) {
    const int tap = step / KS, ks = step % KS;
    const int ky = tap / 3, kx = tap % 3;
    const int chunk = G::chunk(ks, lh);                           // hi chunk of channels [16 ks + 8 lh, +8); lo = chunk + HB
    bf16x8 ah[MBW], al[MBW];
#pragma unroll
    for (int m = 0; m < MBW; ++m) {
      const int iy = y0[m] + ky, ix = x0[m] + kx;
      const int q = ((unsigned)iy < (unsigned)HIN && (unsigned)ix < (unsigned)HIN) ? qb[m] + iy * HIN + ix : ZQ;
      const int sw = G::swz(q);
      ah[m] = *(const bf16x8*)(img + q * PXB + ((chunk ^ sw) * 16));
      al[m] = *(const bf16x8*)(img + q * PXB + (((chunk + G::HB) ^ sw) * 16));
    }
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
      for (int n = 0; n < NBW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l[n], ah[m], acc[m][n], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
      for (int n = 0; n < NBW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h[n], al[m], acc[m][n], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
      for (int n = 0; n < NBW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h[n], ah[m], acc[m][n], 0, 0, 0);
  };
  // register ring of weight fragments, WD K steps deep: the loads of step + WD - 1 are issued before the MFMAs of step.  The
  // loop body covers UNR steps (a multiple of WD, so the ring indices are compile-time constants: registers, not scratch)
  // and is NOT unrolled further: 72 unrolled steps of the 128-channel layer overflow the instruction cache.
  constexpr int WD = KS <= 2 ? 3 : 4, UNR = KS == 1 ? 9 : (KS == 2 ? 6 : KS);      // (a ring 6-8 steps deep was measured: 10-40 % slower on every layer)
  static_assert(STEPS % UNR == 0 && UNR % WD == 0, "ring geometry");
  bf16x8 rh[WD][NBW], rl[WD][NBW];
#pragma unroll
  for (int s0 = 0; s0 < WD - 1; ++s0) load_w(s0, rh[s0], rl[s0]);
#pragma unroll 1
  for (int s0 = 0; s0 < STEPS; s0 += UNR) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int step = s0 + u;
      if (step + WD - 1 < STEPS) load_w(step + WD - 1, rh[(u + WD - 1) % WD], rl[(u + WD - 1) % WD]);
      compute(step, rh[u % WD], rl[u % WD]);
    }
  }
  __syncthreads();                                                // every wave is done with the input image
  stamp(2);

  // ---- raw convolution output (+ bias) -> LDS [pixel][COUT], then the FRN block
#pragma unroll
  for (int m = 0; m < MBW; ++m)
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
      const int pg = (mb0 + m) * 32 + li, pix = pg % G::NPIX, c0 = (nb0 + n) * 32 + 4 * lh;
      float* xb = lds + (pg / G::NPIX) * G::FRN_FLOATS;            // this patch's FRN arrays
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 b4 = *(const float4*)(bias + c0 + 8 * gq);
        *(float4*)(xb + frn_slot(pix, c0 + 8 * gq, COUT)) =
            make_float4(acc[m][n][4 * gq] + b4.x, acc[m][n][4 * gq + 1] + b4.y, acc[m][n][4 * gq + 2] + b4.z, acc[m][n][4 * gq + 3] + b4.w);
      }
    }
  __syncthreads();
  stamp(3);
#pragma unroll
  for (int pp = 0; pp < PP; ++pp)
    if (pp < valid)                                              // workgroup-uniform
      frn_block_body<COUT, HOUT, NT>(lds + pp * G::FRN_FLOATS, fw, fb, eps, g, tau, y, ysp, ldsp, (patch + pp) * G::NPIX, first.prof, stage_gates ? gl : nullptr);
}

// ---------------------------------------------------------------------------------------------- the same block, TWO workgroups per CU (round 5)
// The 32 x 32 layers of ch_conv_block_kernel hold a whole patch twice over in LDS -- the split-bf16 input image (128 KB) and, over it, the raw
// convolution output (128 KB as f32) -- so ONE workgroup fits a CU and its phases (patch load at the HBM rate, matrix phase, statistic / gate /
// apply passes at LDS latency, store) run back to back with nothing under them: a patch costs 64 k cycles of CU time for 19 k of matrix time.
// Here the patch goes through LDS in two HALVES of 16 (stride 2: 8) output rows each: input rows r0 .. r0 + 16 of the half (one halo row re-read
// from L2), the half's accumulators stay in registers while the second half's input replaces the first in LDS, and the FRN (+ CoordAtt) + TLU
// block runs on half images as well -- the accumulators are dumped a half at a time, once for the statistics and pools and once more (from the
// registers they never left) for the apply pass.  <= 76 KB of LDS and <= 128 registers: two workgroups per CU, each other's load and store
// phases under the other's arithmetic.  Same products in the same order per output pixel as ch_conv_block_kernel (K steps ascending, the three
// passes per step in the same order); the statistic sums run over the same pixels in a different association (two halves), i.e. the outputs
// agree with the one-workgroup kernel to f32 rounding of the FRN statistic, not bit for bit.
template <int CIN, int COUT, int STRIDE>
struct HalfGeom {
  static constexpr int HIN = 32, NT = 512, WAVES = 8;
  static constexpr int HOUT = (HIN - 1) / STRIDE + 1, NPIX = HOUT * HOUT, HPIX = NPIX / 2, HROWS = HOUT / 2;     // output pixels / rows per half
  static constexpr int IH = 17;                                        // input rows per half: r0 = 0 (rows 0..16) and r0 = 15 (rows 15..31), both strides
  static constexpr int PXB = CIN * 4, CPP = PXB / 16, ZQ = IH * HIN;   // bytes / chunks per pixel record, index of the all-zero record
  static constexpr int IN_BYTES = (IH * HIN + 1) * PXB;
  static constexpr int MBH = HPIX / 32, NB = COUT / 32, KS = CIN / 16;
  static constexpr int TW = MBH * NB / WAVES;                          // tiles per wave and half (32 -> 32: 2, 32 -> 64 stride 2: 1)
  static_assert(MBH * NB == WAVES * TW && (WAVES / NB) * TW == MBH, "tiles divide over the waves");
  static constexpr int GRP = NT / COUT;
  // FRN arrays (floats): xbh [HPIX][COUT] | red [GRP][COUT] | sc [COUT] | ph [HOUT][COUT] | pw [HOUT][COUT]; the gate weights and `mid` lie over xbh
  static constexpr int XBH = HPIX * COUT, RED = GRP * COUT, FRN_FLOATS = XBH + RED + COUT + 2 * HOUT * COUT;
  static constexpr int GATE_FLOATS = 26 * COUT + 8 + 16 * HOUT;
  static_assert(GATE_FLOATS <= XBH, "gate weights + mid fit over the half image");
  static constexpr int LDS_BYTES = IN_BYTES > FRN_FLOATS * 4 ? IN_BYTES : FRN_FLOATS * 4;
  static constexpr int HB = (CIN < 32 ? CIN : 32) / 8;
  __device__ static __forceinline__ int swz(int q) { return CIN == 16 ? ((q >> 2) & 3) : (CIN == 32 ? ((q >> 1) & 7) : (q & 15)); }
  __device__ static __forceinline__ int chunk(int ks, int lh) { return (ks / (HB / 2)) * (2 * HB) + 2 * (ks % (HB / 2)) + lh; }
};

template <int CIN, int COUT, int STRIDE>
__global__ __launch_bounds__(512, 4) void ch_conv_block_half_kernel(const uint16_t* __restrict__ xin, int64_t ldx, const uint16_t* __restrict__ wpk,
                                                                    const float* __restrict__ bias, const float* __restrict__ fw, const float* __restrict__ fb,
                                                                    float eps, ChGateW g, const float* __restrict__ tau, float* __restrict__ y,
                                                                    uint16_t* __restrict__ ysp, int64_t ldsp, int stagger, int first_wave,
                                                                    unsigned long long* prof) {
  using G = HalfGeom<CIN, COUT, STRIDE>;
  constexpr int NT = G::NT, HIN = G::HIN, ZQ = G::ZQ, PXB = G::PXB, CPP = G::CPP, HOUT = G::HOUT, HPIX = G::HPIX, HROWS = G::HROWS, TW = G::TW, KS = G::KS,
                NB = G::NB, C = COUT, QPP = C / 4, GRP = G::GRP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* img = (char*)lds;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int64_t patch = blockIdx.x;
  ch_stagger(stagger, first_wave);
  auto stamp = [&](int k) __attribute__((always_inline)) { if (prof && blockIdx.x == gridDim.x / 2 && t == 0) prof[k] = __builtin_readcyclecounter(); };
  stamp(0);
  const uint16_t* src = xin + patch * (HIN * HIN) * ldx;

  // this wave's tiles: n-block nb0, local m-blocks mbl0 .. mbl0 + TW - 1 of either half
  const int nb0 = wave % NB, mbl0 = (wave / NB) * TW;
  f32x16 acc[2][TW];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int m = 0; m < TW; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[h][m][r] = 0.f;
  const uint16_t* wl = wpk + lane * 8;
  auto wfrag = [&](int step, int plane) __attribute__((always_inline)) {
    return *(const bf16x8*)(wl + (((int64_t)step * NB + nb0) * 2 + plane) * 512);
  };
  constexpr int STEPS = 9 * KS;
  constexpr int WD = KS <= 2 ? 3 : 4, UNR = KS == 1 ? 9 : (KS == 2 ? 6 : KS);
  static_assert(STEPS % UNR == 0 && UNR % WD == 0, "ring geometry");

#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r0 = h == 0 ? 0 : 15;                               // first input row of the half's LDS image
    if (h == 1) __syncthreads();                                  // every wave is done with the first half's image
    if (t < CPP) *(uint4*)(img + ZQ * PXB + t * 16) = make_uint4(0u, 0u, 0u, 0u);
    {
      constexpr int TOT = G::IH * HIN * CPP;                      // 17 rows x 32 pixels x chunks: not a multiple of LPT * NT -> guarded tail
      constexpr int LPT = 8;
#pragma unroll
      for (int i0 = t; i0 < TOT; i0 += LPT * NT) {
        uint4 v[LPT];
#pragma unroll
        for (int u = 0; u < LPT; ++u) {
          const int i = i0 + u * NT;
          const int ic = i < TOT ? i : TOT - 1;
          v[u] = *(const uint4*)(src + (int64_t)(r0 * HIN + ic / CPP) * ldx + (ic % CPP) * 8);
        }
#pragma unroll
        for (int u = 0; u < LPT; ++u) {
          const int i = i0 + u * NT;
          if (i < TOT) {
            const int q = i / CPP, ch = i % CPP;
            *(uint4*)(img + q * PXB + ((ch ^ G::swz(q)) * 16)) = v[u];
          }
        }
      }
    }
    __syncthreads();
    stamp(1 + 2 * h);
    // input coordinates of tap (0, 0) of this lane's output pixels: global row (may be -1) and column
    int y0[TW], x0[TW];
#pragma unroll
    for (int m = 0; m < TW; ++m) {
      const int p = h * HPIX + (mbl0 + m) * 32 + li;
      y0[m] = (p / HOUT) * STRIDE - 1;
      x0[m] = (p % HOUT) * STRIDE - 1;
    }
    auto compute = [&](int step, const bf16x8& wh, const bf16x8& wlo) __attribute__((always_inline)) {
      const int tap = step / KS, ks = step % KS;
      const int ky = tap / 3, kx = tap % 3;
      const int chunk = G::chunk(ks, lh);
      bf16x8 ah[TW], al[TW];
#pragma unroll
      for (int m = 0; m < TW; ++m) {
        const int iy = y0[m] + ky, ix = x0[m] + kx;
        const int q = ((unsigned)iy < (unsigned)HIN && (unsigned)ix < (unsigned)HIN) ? (iy - r0) * HIN + ix : ZQ;
        const int sw = G::swz(q);
        ah[m] = *(const bf16x8*)(img + q * PXB + ((chunk ^ sw) * 16));
        al[m] = *(const bf16x8*)(img + q * PXB + (((chunk + G::HB) ^ sw) * 16));
      }
#pragma unroll
      for (int m = 0; m < TW; ++m) acc[h][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, ah[m], acc[h][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < TW; ++m) acc[h][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al[m], acc[h][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < TW; ++m) acc[h][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah[m], acc[h][m], 0, 0, 0);
    };
    bf16x8 rh[WD], rl[WD];
#pragma unroll
    for (int s0 = 0; s0 < WD - 1; ++s0) { rh[s0] = wfrag(s0, 0); rl[s0] = wfrag(s0, 1); }
#pragma unroll 1
    for (int s0 = 0; s0 < STEPS; s0 += UNR) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int step = s0 + u;
        if (step + WD - 1 < STEPS) { rh[(u + WD - 1) % WD] = wfrag(step + WD - 1, 0); rl[(u + WD - 1) % WD] = wfrag(step + WD - 1, 1); }
        compute(step, rh[u % WD], rl[u % WD]);
      }
    }
    stamp(2 + 2 * h);
  }
  {                                                               // + bias, once (the accumulators are dumped twice)
    const int c0 = nb0 * 32 + 4 * lh;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const float4 b4 = *(const float4*)(bias + c0 + 8 * gq);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int m = 0; m < TW; ++m) { acc[h][m][4 * gq] += b4.x; acc[h][m][4 * gq + 1] += b4.y; acc[h][m][4 * gq + 2] += b4.z; acc[h][m][4 * gq + 3] += b4.w; }
    }
  }
  __syncthreads();                                                // the input image is dead: the FRN arrays take its place

  // ---------------- FRN (+ CoordAtt) + TLU on half images
  float* xbh = lds;                       // [HPIX][C], quads swizzled by the local pixel index (frn_slot)
  float* red = xbh + G::XBH;              // [GRP][C]
  float* sc = red + G::RED;               // [C]
  float* ph = sc + C;                     // [HOUT][C]: raw row means -> FRN-mapped -> a_h
  float* pw = ph + HOUT * C;              // [HOUT][C]: raw column sums -> FRN-mapped means -> a_w
  float* gl = xbh;                        // gate weights [w1 8C][b1 8][wh 8C][ww 8C][bh C][bw C] and mid [2 HOUT][8]: over xbh between the two dump rounds
  float* mid = gl + 26 * C + 8;
  const bool coord = g.w1 != nullptr;
  // gate weights: requested now, written to LDS after the statistics (their latency runs under the two dump rounds)
  float gv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (coord) {
    if (t < 8 * C) { gv[0] = g.w1[t]; gv[1] = g.wh[t]; gv[2] = g.ww[t]; }
    if (t < C) { gv[3] = g.bh[t]; gv[4] = g.bw[t]; }
    if (t < 8) gv[5] = g.b1[t];
  }
  auto dump = [&](int h) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < TW; ++m) {
      const int pl = (mbl0 + m) * 32 + li, c0 = nb0 * 32 + 4 * lh;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *(float4*)(xbh + frn_slot(pl, c0 + 8 * gq, C)) = make_float4(acc[h][m][4 * gq], acc[h][m][4 * gq + 1], acc[h][m][4 * gq + 2], acc[h][m][4 * gq + 3]);
    }
  };
  float ssq = 0.f;                        // this thread's share of the FRN statistic: channel t % C, pixels t / C + k GRP of both halves
  float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);      // (column item of this thread, summed over the two halves)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    dump(h);
    __syncthreads();
    {
      const int c = t % C, gq = t / C;
      float s4[4] = {0.f, 0.f, 0.f, 0.f};
      static_assert((HPIX / GRP) % 4 == 0, "four partial sums per thread");
#pragma unroll 2
      for (int pix = gq; pix < HPIX; pix += 4 * GRP) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const float v = xbh[frn_slot(pix + u * GRP, c, C)]; s4[u] = fmaf(v, v, s4[u]); }
      }
      ssq += (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    if (coord) {
      // items: [0, HROWS QPP): row means of this half's rows; [HROWS QPP, (HROWS + HOUT) QPP): column sums over this half's rows
      static_assert((HROWS + HOUT) * QPP <= NT, "one pool item per thread");
      if (t < (HROWS + HOUT) * QPP) {
        const bool over_x = t < HROWS * QPP;
        const int j = over_x ? t : t - HROWS * QPP, line = j / QPP, q = j % QPP;
        constexpr int NK = HOUT;                                  // row length; a column holds HROWS = NK / 2 pixels of this half
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        const int nk = over_x ? NK : HROWS;
        for (int k = 0; k < nk; k += 2) {
          const int p0 = over_x ? line * HOUT + k : k * HOUT + line, p1 = over_x ? p0 + 1 : p0 + HOUT;
          const float4 v0 = *(const float4*)(xbh + p0 * C + ((q ^ (p0 & 7)) << 2)), v1 = *(const float4*)(xbh + p1 * C + ((q ^ (p1 & 7)) << 2));
          a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
          a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        }
        if (over_x) *(float4*)(ph + (h * HROWS + line) * C + 4 * q) = make_float4((a0.x + a1.x) / (float)HOUT, (a0.y + a1.y) / (float)HOUT,
                                                                                 (a0.z + a1.z) / (float)HOUT, (a0.w + a1.w) / (float)HOUT);
        else { colsum.x += a0.x + a1.x; colsum.y += a0.y + a1.y; colsum.z += a0.z + a1.z; colsum.w += a0.w + a1.w; }
      }
    }
    __syncthreads();                                              // the half image is read; the next dump may overwrite it
  }
  stamp(5);
  red[(t / C) * C + t % C] = ssq;
  if (coord && t >= HROWS * QPP && t < (HROWS + HOUT) * QPP) {
    const int j = t - HROWS * QPP;
    *(float4*)(pw + (j / QPP) * C + 4 * (j % QPP)) = make_float4(colsum.x / (float)HOUT, colsum.y / (float)HOUT, colsum.z / (float)HOUT, colsum.w / (float)HOUT);
  }
  if (coord) {                                                    // gate weights over the (dead) half image
    if (t < 8 * C) { gl[t] = gv[0]; gl[8 * C + 8 + t] = gv[1]; gl[16 * C + 8 + t] = gv[2]; }
    if (t < C) { gl[24 * C + 8 + t] = gv[3]; gl[25 * C + 8 + t] = gv[4]; }
    if (t < 8) gl[8 * C + t] = gv[5];
  }
  __syncthreads();
  if (t < C) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < GRP; ++q) s += red[q * C + t];
    sc[t] = fw[t] * rsqrtf(s / (float)G::NPIX + eps);
  }
  __syncthreads();
  if (coord) {
    const float *g_w1 = gl, *g_b1 = gl + 8 * C, *g_wh = gl + 8 * C + 8, *g_ww = gl + 16 * C + 8, *g_bh = gl + 24 * C + 8, *g_bw = gl + 25 * C + 8;
    // pools of the FRN output = FRN affine map of the raw pools
    for (int i = t; i < 2 * HOUT * C; i += NT) ph[i] = fmaf(ph[i], sc[i % C], fb[i % C]);       // (ph and pw are contiguous)
    __syncthreads();
    for (int i = t; i < 2 * HOUT * 8; i += NT) {
      const int r = i >> 3, m = i & 7;
      const float* srcp = ph + r * C;                             // rows 0 .. HOUT - 1: ph, HOUT .. 2 HOUT - 1: pw
      float a4[4] = {g_b1[m], 0.f, 0.f, 0.f};
#pragma unroll 8
      for (int k = 0; k < C; k += 4) {
        const float4 wv = *(const float4*)(g_w1 + m * C + k), xv = *(const float4*)(srcp + k);
        a4[0] = fmaf(xv.x, wv.x, a4[0]); a4[1] = fmaf(xv.y, wv.y, a4[1]); a4[2] = fmaf(xv.z, wv.z, a4[2]); a4[3] = fmaf(xv.w, wv.w, a4[3]);
      }
      const float av = (a4[0] + a4[1]) + (a4[2] + a4[3]);
      mid[i] = av * (fminf(fmaxf(av + 3.f, 0.f), 6.f) / 6.f);
    }
    __syncthreads();
    for (int i = t; i < 2 * HOUT * C; i += NT) {
      const bool is_h = i < HOUT * C;
      const int j = is_h ? i : i - HOUT * C, r = j / C, ch = j % C;
      const float* wt = (is_h ? g_wh : g_ww) + ch * 8;
      const float* mr = mid + (is_h ? r : HOUT + r) * 8;
      const float4 w0 = *(const float4*)wt, w1v = *(const float4*)(wt + 4), m0 = *(const float4*)mr, m1 = *(const float4*)(mr + 4);
      float av = (is_h ? g_bh : g_bw)[ch];
      av = fmaf(m0.x, w0.x, av); av = fmaf(m0.y, w0.y, av); av = fmaf(m0.z, w0.z, av); av = fmaf(m0.w, w0.w, av);
      av = fmaf(m1.x, w1v.x, av); av = fmaf(m1.y, w1v.y, av); av = fmaf(m1.z, w1v.z, av); av = fmaf(m1.w, w1v.w, av);
      ph[i] = 1.f / (1.f + __expf(-av));
    }
    __syncthreads();                                              // gates final; the gate weights over xbh are dead
  }
  stamp(6);
  const int64_t pbase = patch * G::NPIX;
  constexpr int NQ = HPIX * QPP / NT;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    dump(h);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int i = t + NT * j, pix = i / QPP, ch = 4 * (i % QPP), yy = h * HROWS + pix / HOUT, xx = pix % HOUT;
      const float4 v = *(const float4*)(xbh + frn_slot(pix, ch, C));
      const float4 s4 = *(const float4*)(sc + ch), b4 = *(const float4*)(fb + ch);
      float r[4] = {fmaf(v.x, s4.x, b4.x), fmaf(v.y, s4.y, b4.y), fmaf(v.z, s4.z, b4.z), fmaf(v.w, s4.w, b4.w)};
      if (coord) {
        const float4 g1 = *(const float4*)(ph + yy * C + ch), g2 = *(const float4*)(pw + xx * C + ch);
        r[0] = r[0] * g2.x * g1.x; r[1] = r[1] * g2.y * g1.y; r[2] = r[2] * g2.z * g1.z; r[3] = r[3] * g2.w * g1.w;
      }
      const float4 t4 = *(const float4*)(tau + ch);
      r[0] = fmaxf(r[0], t4.x); r[1] = fmaxf(r[1], t4.y); r[2] = fmaxf(r[2], t4.z); r[3] = fmaxf(r[3], t4.w);
      const int64_t gp = pbase + h * HPIX + pix;
      if (y) *(float4*)(y + gp * C + ch) = make_float4(r[0], r[1], r[2], r[3]);
      if (ysp) store_split4(ysp + gp * ldsp + spl_col(ch), r);
    }
    if (h == 0) __syncthreads();
  }
  stamp(7);
}

// ---------------------------------------------------------------------------------------------- fused SandGlass block
// One workgroup per patch, the whole block of models.py:182-235 (+ the outer residual of 383-389) with the patch's activation
// resident in LDS (H * W * C = 32768 floats = 128 KiB: 32x32x32 or 16x16x64):
//   0  x -> LDS (bulk, coalesced) and this thread's own quads -> registers (for the residual)
//   A  y = ReLU6(dw3x3(x) + BN)                       from the LDS copy into registers, then over the copy
//   A2 pools of y over W and over H                   LDS -> LDS
//   B  CoordAtt gate MLP: 8-channel bottleneck, sigmoid gates a_h, a_w (they overwrite the pools)
//   C  z = ReLU6(W1 (W0 (y a_w a_h) + b0) + b1)      per pixel, in place
//   D  out = 2 x + dw3x3(z) + BN                      z from LDS, x from registers, out as SPL32 split-bf16 pixel rows
// LDS image of y / z: [pixel][C] with the 16-byte quads of a pixel XOR-swizzled by the pixel index, so that both the
// quad-per-lane passes (A, D) and the pixel-per-lane pass (C) are bank-conflict free without padding.
struct ChSandglassW {          // all f32, BatchNorm folded
  const float* dw0;   // [9][C]
  const float* dw0b;  // [C]
  const float* w1;    // [8][C]   CoordAtt conv1 (+bn1)
  const float* b1;    // [8]
  const float* wh;    // [C][8]
  const float* bh;    // [C]
  const float* ww;    // [C][8]
  const float* bw;    // [C]
  const float* p0;    // [16][C]  pw-linear (+BN)
  const float* p0b;   // [16]
  const float* p1;    // [C][16]  pw (+BN), ReLU6 after
  const float* p1b;   // [C]
  const float* dw1;   // [9][C]
  const float* dw1b;  // [C]
  unsigned long long* prof;   // GIMS_CH_PROF=1: cycle stamps of one workgroup, else null
  int stagger, first_wave;    // GIMS_CH_STAGGER (ch_stagger)
};

template <int C, int HW, int NT>      // HW = H = W, NT threads
__global__ __launch_bounds__(NT) void ch_sandglass_kernel(const float* __restrict__ x, ChSandglassW wts, uint16_t* __restrict__ out, int64_t ldo,
                                                          int n_patches) {
  constexpr int NPIX = HW * HW, QPP = C / 4, MASK = QPP - 1;          // quads per pixel
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ybuf = lds;                        // [NPIX][C] swizzled
  float* ph = ybuf + NPIX * C;              // [HW][C]  -> a_h
  float* pw = ph + HW * C;                  // [HW][C]  -> a_w
  float* mid = pw + HW * C;                 // [2 HW][8]
  float* wl = mid + 2 * HW * 8;             // dw0 [9][C], dw0b [C], dw1 [9][C], dw1b [C], p0 [16][C], p0b [16], p1 [C][16], p1b [C]
  constexpr int P0P = C + 4, P1P = 20;      // padded pitches of the pointwise weights: their MFMA fragment reads (one row per lane) are conflict-free
  float* l_dw0 = wl, *l_dw0b = l_dw0 + 9 * C, *l_dw1 = l_dw0b + C, *l_dw1b = l_dw1 + 9 * C, *l_p0 = l_dw1b + C, *l_p0b = l_p0 + 16 * P0P,
        *l_p1 = l_p0b + 16, *l_p1b = l_p1 + C * P1P;
  // gate weights of CoordAtt (w1 [8][C], b1 [8], wh / ww [C][8], bh / bw [C]): read from global memory inside phase B they cost two L2 round
  // trips per patch in the middle of the kernel; staged here, their latency hides under the patch load
  float* l_w1 = l_p1b + C, *l_b1 = l_w1 + 8 * C, *l_wh = l_b1 + 8, *l_ww = l_wh + 8 * C, *l_bh = l_ww + 8 * C, *l_bw = l_bh + C;
  const int t = threadIdx.x;
  auto stamp = [&](int k) __attribute__((always_inline)) { if (wts.prof && blockIdx.x == gridDim.x / 2 && t == 0) wts.prof[k] = __builtin_readcyclecounter(); };
  ch_stagger(wts.stagger, wts.first_wave);
  stamp(0);
  // ---- 0: the patch is REQUESTED first (NQ independent, fully coalesced 16-byte loads per thread), the weights are staged into LDS while
  // it is in flight: one memory latency for both.  x is NOT kept in registers for the final residual: next to the accumulators and taps of
  // the depthwise passes its 4 NQ registers spilled to scratch; phase D requests it again (an L2 / Infinity-Cache hit a few microseconds later)
  // before its own arithmetic
  // Thread <-> data: channel quad cq of the pixels (xx, y0 .. y0 + NQ - 1) -- a VERTICAL strip.  A wave covers XW adjacent columns x all
  // channel quads = one contiguous KiB per image row (coalesced loads and stores), and the depthwise passes read every input row of
  // the strip ONCE for the three output rows it feeds: (NQ + 2) x 3 LDS reads per pass instead of NQ x 9 (these passes are
  // LDS-bandwidth bound).
  constexpr int NQ = NPIX * QPP / NT, XW = 64 / QPP, XG = HW / XW;
  static_assert((NT / 64 / XG) * NQ == HW && XG * XW == HW, "the strips tile the patch");
  const int cq = (t & 63) % QPP, xx = ((t >> 6) % XG) * XW + (t & 63) / QPP, y0 = __builtin_amdgcn_readfirstlane(((t >> 6) / XG) * NQ);
  // PERSISTENT: a workgroup walks patches blockIdx.x, + gridDim.x, ... -- the weights are staged once per workgroup instead of once per
  // patch, and a CU does not pay a workgroup dispatch (16 waves, 150 KB of LDS) between two patches.
  f32x4 xq[NQ];
  {
    const float* xp = x + (int64_t)blockIdx.x * NPIX * C;
#pragma unroll
    for (int j = 0; j < NQ; ++j) xq[j] = *(const f32x4*)(xp + (int64_t)((y0 + j) * HW + xx) * C + 4 * cq);
  }
  for (int i = t; i < 9 * C; i += NT) { l_dw0[i] = wts.dw0[i]; l_dw1[i] = wts.dw1[i]; }
  for (int i = t; i < 16 * C; i += NT) { l_p0[(i / C) * P0P + i % C] = wts.p0[i]; l_p1[(i / 16) * P1P + i % 16] = wts.p1[i]; }
  if (t < C) { l_dw0b[t] = wts.dw0b[t]; l_dw1b[t] = wts.dw1b[t]; l_p1b[t] = wts.p1b[t]; }
  if (t < 16) l_p0b[t] = wts.p0b[t];
  for (int i = t; i < 8 * C; i += NT) { l_w1[i] = wts.w1[i]; l_wh[i] = wts.wh[i]; l_ww[i] = wts.ww[i]; }
  if (t < C) { l_bh[t] = wts.bh[t]; l_bw[t] = wts.bw[t]; }
  if (t < 8) l_b1[t] = wts.b1[t];
  stamp(1);
  auto slot = [&](int pix, int cq) { return pix * C + 4 * (cq ^ (pix & MASK)); };

  // (32x32x32 only: at 16x16x64 the looped form measured slower, 495 -> 556 us per 8192 patches -- it is launched one workgroup per patch and
  // compiled without the loop)
  constexpr bool PERSIST = C == 32;
  int patch = blockIdx.x;
  do {
  // every thread-derived index is re-derived from an OPAQUE copy of the thread id inside the loop: as loop invariants the compiler hoisted the
  // weight fragments, taps and addresses of all phases out of the patch loop and spilled 100 registers
  int t_opaque = threadIdx.x;
  if (PERSIST) asm volatile("" : "+v"(t_opaque));
  const int t = t_opaque;
  const int cq = (t & 63) % QPP, xx = ((t >> 6) % XG) * XW + (t & 63) / QPP, y0 = __builtin_amdgcn_readfirstlane(((t >> 6) / XG) * NQ);
  const int64_t pbase = (int64_t)patch * NPIX;
  const float* xp = x + pbase * C;
  if (patch != (int)blockIdx.x) {
    __syncthreads();                                        // the previous patch's last LDS reads (phase D) are done
#pragma unroll
    for (int j = 0; j < NQ; ++j) xq[j] = *(const f32x4*)(xp + (int64_t)((y0 + j) * HW + xx) * C + 4 * cq);
  }
#pragma unroll
  for (int j = 0; j < NQ; ++j) *(f32x4*)(ybuf + slot((y0 + j) * HW + xx, cq)) = xq[j];
  asm volatile("" ::: "memory");
  __syncthreads();
  stamp(2);
  // depthwise 3x3 of the LDS image for this thread's strip: acc[j] (preset to the bias) += taps in the reference's order (row-major
  // over the 3x3 window; out-of-image taps add nothing).  The quad swizzle depends on the column only (HW is a multiple of QPP).
  static_assert(HW % QPP == 0 || QPP % HW == 0, "swizzle independent of the row");
  auto dw3x3 = [&](const f32x4 (&kw)[9], f32x4 (&acc)[NQ], auto o_first, auto o_count) __attribute__((always_inline)) {   // output rows [O0, O0 + ON) of the strip
    constexpr int O0 = decltype(o_first)::value, ON = decltype(o_count)::value;
    int xo[3];
    float keep[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int x2 = xx + d - 1, xc = x2 < 0 ? 0 : (x2 >= HW ? HW - 1 : x2);
      xo[d] = xc * C + 4 * (cq ^ (xc & MASK));
      keep[d] = (x2 >= 0 && x2 < HW) ? 1.f : 0.f;
    }
#pragma unroll
    for (int rr = O0 - 1; rr <= O0 + ON; ++rr) {
      const int yin = y0 + rr;
      if (yin < 0 || yin >= HW) continue;                  // wave-uniform
      f32x4 v[3];                                         // ext-vector arithmetic: the compiler pairs it into v_pk_mul / v_pk_fma_f32 (these passes are VALU-issue bound)
#pragma unroll
      for (int d = 0; d < 3; ++d) v[d] = *(const f32x4*)(ybuf + yin * HW * C + xo[d]) * keep[d];
#pragma unroll
      for (int dy = 1; dy >= -1; --dy) {                    // output row rr - dy takes this input row as window row dy + 1
        const int o = rr - dy;
        if (o < O0 || o >= O0 + ON) continue;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          acc[o] = __builtin_elementwise_fma(v[d], kw[(dy + 1) * 3 + d], acc[o]);
        }
      }
      if (rr & 1) asm volatile("" ::: "memory");            // at most two rows of reads in flight: hoisting all of them spilled registers
    }
  };
  // ---- A: depthwise 3x3 + BN + ReLU6 from the LDS copy into registers, then over the copy.  The loop is unrolled (register
  // arrays), but the thread index is re-derived from an opaque copy every iteration: as loop invariants the compiler hoisted
  // all NQ x 9 addresses and weights and spilled 900 registers.
  // a thread's quads all belong to ONE channel quad (NT % QPP == 0): its nine tap weights stay in registers (they were half of
  // the LDS reads of this LDS-bandwidth-bound pass)
  f32x4 kw[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) kw[tap] = *(const f32x4*)(l_dw0 + tap * C + 4 * cq);
  const f32x4 kb0 = *(const f32x4*)(l_dw0b + 4 * cq);
  f32x4 yq[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) yq[j] = kb0;
  dw3x3(kw, yq, std::integral_constant<int, 0>{}, std::integral_constant<int, NQ>{});
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
#pragma unroll
    for (int e = 0; e < 4; ++e) yq[j][e] = fminf(fmaxf(yq[j][e], 0.f), 6.f);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NQ; ++j) *(f32x4*)(ybuf + slot((y0 + j) * HW + xx, cq)) = yq[j];
  __syncthreads();
  stamp(3);
  // ---- A2: pools (mean over x for every row, mean over y for every column): one (line, channel quad) per thread, HW float4 reads
  for (int i = t; i < 2 * HW * QPP; i += NT) {
    const bool over_x = i < HW * QPP;
    const int j = over_x ? i : i - HW * QPP, line = j / QPP, q = j % QPP;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#pragma unroll
    for (int k = 0; k < HW; k += 2) {
      const int p0 = over_x ? line * HW + k : k * HW + line, p1 = over_x ? p0 + 1 : p0 + HW;
      const float4 v0 = *(const float4*)(ybuf + slot(p0, q)), v1 = *(const float4*)(ybuf + slot(p1, q));
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
    }
    *(float4*)((over_x ? ph : pw) + line * C + 4 * q) =
        make_float4((a0.x + a1.x) / (float)HW, (a0.y + a1.y) / (float)HW, (a0.z + a1.z) / (float)HW, (a0.w + a1.w) / (float)HW);
  }
  __syncthreads();
  stamp(4);
  // ---- B: gate MLP
  for (int i = t; i < 2 * HW * 8; i += NT) {
    const int r = i >> 3, m = i & 7;
    const float* src = r < HW ? ph + r * C : pw + (r - HW) * C;
    float a4[4] = {l_b1[m], 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int k = 0; k < C; k += 4) {
      const float4 wv = *(const float4*)(l_w1 + m * C + k), xv = *(const float4*)(src + k);
      a4[0] = fmaf(xv.x, wv.x, a4[0]); a4[1] = fmaf(xv.y, wv.y, a4[1]); a4[2] = fmaf(xv.z, wv.z, a4[2]); a4[3] = fmaf(xv.w, wv.w, a4[3]);
    }
    const float acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    mid[i] = acc * (fminf(fmaxf(acc + 3.f, 0.f), 6.f) / 6.f);
  }
  __syncthreads();
  for (int i = t; i < 2 * HW * C; i += NT) {
    const bool is_h = i < HW * C;
    const int j = is_h ? i : i - HW * C, r = j / C, ch = j % C;
    const float* wt = (is_h ? l_wh : l_ww) + ch * 8;
    const float* mr = mid + (is_h ? r : HW + r) * 8;
    float acc = (is_h ? l_bh : l_bw)[ch];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc = fmaf(mr[m], wt[m], acc);
    // a_w is stored with its channel quads XOR-swizzled by the column index: phase C reads one column per lane (stride C floats)
    (is_h ? ph : pw)[is_h ? j : r * C + 4 * ((ch >> 2) ^ (r & MASK)) + (ch & 3)] = 1.f / (1.f + __expf(-acc));
  }
  __syncthreads();
  stamp(5);
  // ---- C: per pixel  z = ReLU6(W1 (W0 (y a_w a_h) + b0) + b1), in place -- on the matrix cores, three bf16 passes per product like the
  // convolutions.  A wave owns 32-pixel blocks, one pixel per lane COLUMN:  hid^T [16 (+16 zero rows)][32 px] = W0 T^T  over C / 16
  // k-steps, then  z^T [32 ch][32 px] = W1 hid  per 32-channel block.  The C layout of hid^T (lane = pixel, registers 0-7 = hidden units
  // (j & 3) + 8 (j >> 2) + 4 lh) IS a B-operand layout of the second product once W1's k-slots are loaded in that order, so the
  // hidden vector never leaves registers; the z^T registers 4g..4g+3 are channel quad 2g + lh of the lane's pixel and go back over the
  // pixel's LDS record as float4.  (As per-pixel VALU work this phase was 1 M fma per patch: 23 k of the kernel's 79 k cycles at
  // 32x32x32, and at 16x16x64 only a quarter of the threads had a pixel.)
  {
    const int lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
    constexpr int KS = C / 16, NBK = C / 32, NWAVES = NT / 64, NBLK = NPIX / 32;
    auto split8 = [&](const float (&v)[8], bf16x8& hi, bf16x8& lo) __attribute__((always_inline)) {
      uint32_t h[4], l[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        h[j] = pack_bf2(v[2 * j], v[2 * j + 1]);
        l[j] = pack_bf2(v[2 * j] - __uint_as_float(h[j] << 16), v[2 * j + 1] - __uint_as_float(h[j] & 0xffff0000u));
      }
      hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
      lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
    };
    bf16x8 a0h[KS], a0l[KS], a1h[NBK], a1l[NBK];
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {                       // W0 rows = hidden units (rows 16-31 are padding), k = channels 16 s + 8 lh ..
      float w[8];
      const float4 w0 = *(const float4*)(l_p0 + (li & 15) * P0P + 16 * s2 + 8 * lh), w1 = *(const float4*)(l_p0 + (li & 15) * P0P + 16 * s2 + 8 * lh + 4);
      const float keep = li < 16 ? 1.f : 0.f;
      w[0] = w0.x * keep; w[1] = w0.y * keep; w[2] = w0.z * keep; w[3] = w0.w * keep;
      w[4] = w1.x * keep; w[5] = w1.y * keep; w[6] = w1.z * keep; w[7] = w1.w * keep;
      split8(w, a0h[s2], a0l[s2]);
    }
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {                      // W1 rows = output channels 32 nb + li, k-slot j = hidden (j & 3) + 8 (j >> 2) + 4 lh
      const float4 w0 = *(const float4*)(l_p1 + (32 * nb + li) * P1P + 4 * lh), w1 = *(const float4*)(l_p1 + (32 * nb + li) * P1P + 8 + 4 * lh);
      const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
      split8(w, a1h[nb], a1l[nb]);
    }
    const float4 hb0 = *(const float4*)(l_p0b + 4 * lh), hb1 = *(const float4*)(l_p0b + 8 + 4 * lh);
    for (int blk = wave; blk < NBLK; blk += NWAVES) {
      int l2 = li;
      asm volatile("" : "+v"(l2));                          // (keeps the per-block addresses out of the loop-invariant set: registers)
      const int pix = 32 * blk + l2, yy = pix / HW, xx = pix % HW;
      f32x16 hacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) {
        const int q0 = 4 * s2 + 2 * lh;
        const float4 v0 = *(const float4*)(ybuf + slot(pix, q0)), v1 = *(const float4*)(ybuf + slot(pix, q0 + 1));
        const float4 g0 = *(const float4*)(ph + yy * C + 4 * q0), g1 = *(const float4*)(ph + yy * C + 4 * q0 + 4);
        const float4 e0 = *(const float4*)(pw + xx * C + 4 * (q0 ^ (xx & MASK))), e1 = *(const float4*)(pw + xx * C + 4 * ((q0 + 1) ^ (xx & MASK)));
        const float tv[8] = {v0.x * e0.x * g0.x, v0.y * e0.y * g0.y, v0.z * e0.z * g0.z, v0.w * e0.w * g0.w,
                             v1.x * e1.x * g1.x, v1.y * e1.y * g1.y, v1.z * e1.z * g1.z, v1.w * e1.w * g1.w};
        bf16x8 bh, bl;
        split8(tv, bh, bl);
        hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l[s2], bh, hacc, 0, 0, 0);      // small terms first
        hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h[s2], bl, hacc, 0, 0, 0);
        hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h[s2], bh, hacc, 0, 0, 0);
      }
      const float hid[8] = {hacc[0] + hb0.x, hacc[1] + hb0.y, hacc[2] + hb0.z, hacc[3] + hb0.w, hacc[4] + hb1.x, hacc[5] + hb1.y, hacc[6] + hb1.z, hacc[7] + hb1.w};
      bf16x8 hh, hl;
      split8(hid, hh, hl);
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l[nb], hh, z, 0, 0, 0);
        z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h[nb], hl, z, 0, 0, 0);
        z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h[nb], hh, z, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int cq = 8 * nb + 2 * g + lh;
          const float4 b1 = *(const float4*)(l_p1b + 4 * cq);
          *(float4*)(ybuf + slot(pix, cq)) = make_float4(fminf(fmaxf(z[4 * g] + b1.x, 0.f), 6.f), fminf(fmaxf(z[4 * g + 1] + b1.y, 0.f), 6.f),
                                                         fminf(fmaxf(z[4 * g + 2] + b1.z, 0.f), 6.f), fminf(fmaxf(z[4 * g + 3] + b1.w, 0.f), 6.f));
        }
      }
    }
  }
  __syncthreads();
  stamp(6);
  // ---- D: out = 2 x + dw3x3(z) + BN, as split-bf16 pixel rows (x from the registers of step 0)
#pragma unroll
  for (int j = 0; j < NQ; ++j) xq[j] = __builtin_nontemporal_load((const f32x4*)(xp + (int64_t)((y0 + j) * HW + xx) * C + 4 * cq));
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) kw[tap] = *(const f32x4*)(l_dw1 + tap * C + 4 * cq);
  const f32x4 kb0d = *(const f32x4*)(l_dw1b + 4 * cq);
#pragma unroll
  for (int j = 0; j < NQ; ++j) yq[j] = kb0d;
  // in two halves when the strip is 8 rows: 16 accumulator registers instead of 32 next to x (32) and the taps (36) -- the whole
  // strip at once spilled 28 registers; the price is two input rows read twice
  constexpr int HALF = NQ > 4 ? NQ / 2 : NQ;
  auto finish = [&](auto o_first) __attribute__((always_inline)) {
    constexpr int O0 = decltype(o_first)::value;
    dw3x3(kw, yq, o_first, std::integral_constant<int, HALF>{});
#pragma unroll
    for (int j = O0; j < O0 + HALF; ++j) {
      const float r[4] = {fmaf(xq[j][0], 2.f, yq[j][0]), fmaf(xq[j][1], 2.f, yq[j][1]), fmaf(xq[j][2], 2.f, yq[j][2]), fmaf(xq[j][3], 2.f, yq[j][3])};
      store_split4(out + (pbase + (y0 + j) * HW + xx) * ldo + spl_col(4 * cq), r);
    }
  };
  finish(std::integral_constant<int, 0>{});
  if constexpr (HALF < NQ) finish(std::integral_constant<int, HALF>{});
  stamp(7);
  patch += gridDim.x;
  } while (PERSIST && patch < n_patches);
}

// one wave per row of `c` (<= 256) values: y = x / sqrt(sum x^2 + eps)
__global__ __launch_bounds__(256) void ch_l2norm_kernel(const float* __restrict__ x, int64_t rows, int c, float eps, float* __restrict__ y) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float v[4], s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int k = lane + 64 * j; v[j] = k < c ? x[row * c + k] : 0.f; s = fmaf(v[j], v[j], s); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float inv = 1.f / sqrtf(s + eps);
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int k = lane + 64 * j; if (k < c) y[row * c + k] = v[j] * inv; }
}

// elementwise clamp to [0, 6] (ReLU6 after a pointwise GEMM) and  y = a + b
__global__ __launch_bounds__(256) void ch_relu6_kernel(float* __restrict__ x, int64_t total) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= total) return;
  float4 v = *(float4*)(x + i4);
  v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f); v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f);
  *(float4*)(x + i4) = v;
}

}  // namespace gims

using namespace gims;

extern "C" int gims_ch_frn_stats(const float* x, int64_t patches, int32_t hw, int32_t c, const float* weight, float eps, float* scale, void* stream) {
  GIMS_CHECK_ARG(x && weight && scale && patches > 0 && hw > 0 && c > 0 && eps >= 0.f, "gims_ch_frn_stats: bad arguments");
  hipLaunchKernelGGL(ch_frn_stats_kernel, dim3((unsigned)patches, (c + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, hw, c, weight, eps, scale);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_ch_pool_hw(const float* x, int64_t patches, int32_t h, int32_t w, int32_t c, const float* s, const float* b, float* ph, float* pw,
                               float* rowsq, void* stream) {
  GIMS_CHECK_ARG(x && ph && pw && patches > 0 && h > 0 && w > 0 && c > 0, "gims_ch_pool_hw: bad arguments");
  const int64_t n = patches * (h > w ? h : w) * c;
  hipLaunchKernelGGL(ch_pool_hw_kernel, dim3((unsigned)((n + 255) / 256), 2), dim3(256), 0, (hipStream_t)stream, x, patches, h, w, c, s, b, ph, pw, rowsq);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_ch_frn_from_rows(const float* rowsq, int64_t patches, int32_t h, int32_t w, int32_t c, const float* weight, float eps, float* scale,
                                     void* stream) {
  GIMS_CHECK_ARG(rowsq && weight && scale && patches > 0 && h > 0 && w > 0 && c > 0 && eps >= 0.f, "gims_ch_frn_from_rows: bad arguments");
  hipLaunchKernelGGL(ch_frn_from_rows_kernel, dim3((unsigned)((patches * c + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rowsq, patches, h, w, c, weight,
                     eps, scale);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_ch_gates(const float* ph, const float* pw, int64_t patches, int32_t h, int32_t w, int32_t c, const float* w1, const float* b1,
                             const float* wh, const float* bh, const float* ww, const float* bw, const float* frn_scale, const float* frn_bias,
                             float* ah, float* aw, void* stream) {
  GIMS_CHECK_ARG(ph && pw && w1 && b1 && wh && bh && ww && bw && ah && aw && patches > 0 && h + w <= 64 && ((frn_scale == nullptr) == (frn_bias == nullptr)),
                 "gims_ch_gates: bad arguments (h + w <= 64)");
  hipLaunchKernelGGL(ch_gates_kernel, dim3((unsigned)patches), dim3(256), 0, (hipStream_t)stream, ph, pw, h, w, c, w1, b1, wh, bh, ww, bw, frn_scale, frn_bias,
                     ah, aw);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_ch_apply(const float* x, int64_t patches, int32_t h, int32_t w, int32_t c, const float* s, const float* b, const float* ah,
                             const float* aw, const float* tau, float* y, uint16_t* y_split, int64_t ld_split, void* stream) {
  GIMS_CHECK_ARG(x && (y || y_split) && patches > 0 && (c % 4) == 0 && ((ah == nullptr) == (aw == nullptr)), "gims_ch_apply: bad arguments (c %% 4 == 0)");
  GIMS_CHECK_ARG(!y_split || ((c % 32) == 0 && ld_split >= 2 * (int64_t)c && (ld_split % 4) == 0), "gims_ch_apply: split output needs c %% 32 == 0, pitch >= 2c");
  const int64_t total = patches * h * w * c;
  hipLaunchKernelGGL(ch_apply_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, total, h, w, c, s, b, ah, aw, tau, y,
                     y_split, ld_split);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_ch_im2col3(const float* x, int64_t patches, int32_t h, int32_t w, int32_t c, int32_t stride, uint16_t* out, int64_t ld,
                               int32_t kpad, void* stream) {
  GIMS_CHECK_ARG(x && out && patches > 0 && (stride == 1 || stride == 2) && (kpad % 32) == 0 && kpad >= 9 * c && ld >= 2 * (int64_t)kpad && (ld % 64) == 0,
                 "gims_ch_im2col3: bad arguments (kpad %% 32 == 0, kpad >= 9c, ld >= 2 kpad, ld %% 64 == 0)");
  const int ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1;
  const int64_t rows = patches * ho * wo, n = rows * (kpad / 4);
  hipLaunchKernelGGL(ch_im2col3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, rows, h, w, c, stride, ho, wo, kpad, out, ld);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_ch_dwconv3(const float* x, int64_t patches, int32_t h, int32_t w, int32_t c, const float* wt, const float* bias, int32_t relu6_out,
                               const float* res, float res_scale, float* y, uint16_t* y_split, int64_t ld_split, void* stream) {
  GIMS_CHECK_ARG(x && (y || y_split) && wt && bias && patches > 0 && (c % 4) == 0, "gims_ch_dwconv3: bad arguments (c %% 4 == 0)");
  GIMS_CHECK_ARG(!y_split || ((c % 32) == 0 && ld_split >= 2 * (int64_t)c && (ld_split % 4) == 0), "gims_ch_dwconv3: split output needs c %% 32 == 0, pitch >= 2c");
  const int64_t total = patches * h * w * c;
  hipLaunchKernelGGL(ch_dwconv3_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, total, h, w, c, wt, bias, relu6_out, res,
                     res_scale, y, y_split, ld_split);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_ch_gate_pw_pw(const float* x, int64_t patches, int32_t h, int32_t w, int32_t c, const float* ah, const float* aw, const float* w0,
                                  const float* b0, const float* w1, const float* b1, float* z, void* stream) {
  GIMS_CHECK_ARG(x && ah && aw && w0 && b0 && w1 && b1 && z && patches > 0 && (c == 32 || c == 64), "gims_ch_gate_pw_pw: bad arguments (c = 32 or 64, hidden 16)");
  const int64_t pixels = patches * h * w;
  const dim3 grid((unsigned)((pixels + 255) / 256));
  if (c == 32) hipLaunchKernelGGL(ch_gate_pw_pw_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, x, pixels, h, w, ah, aw, w0, b0, w1, b1, z);
  else hipLaunchKernelGGL(ch_gate_pw_pw_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, x, pixels, h, w, ah, aw, w0, b0, w1, b1, z);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_ch_input_block(const float* patches, int64_t n, const float* frn_weight, const float* frn_bias, float eps, const float* tau,
                                   uint16_t* out, int64_t ld, void* stream) {
  GIMS_CHECK_ARG(patches && frn_weight && frn_bias && tau && out && n > 0 && eps >= 0.f && ld >= 128 && (ld % 64) == 0 && (((uintptr_t)out) & 15) == 0,
                 "gims_ch_input_block: bad arguments (SPL32 rows of K = 64: pitch >= 128, %% 64 == 0)");
  hipLaunchKernelGGL(ch_input_block_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, patches, frn_weight, frn_bias, eps, tau, out, ld);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_ch_frn_block(const float* x, int64_t patches, int32_t hw, int32_t c, const float* frn_weight, const float* frn_bias, float eps,
                                 const float* const* gate_w /* 6 device pointers (w1, b1, wh, bh, ww, bw) or NULL: no CoordAtt */, const float* tau,
                                 float* y, uint16_t* y_split, int64_t ld_split, void* stream) {
  GIMS_CHECK_ARG(x && frn_weight && frn_bias && tau && (y || y_split) && patches > 0 && eps >= 0.f &&
                     ((c == 32 && hw == 32) || (c == 64 && hw == 16) || (c == 128 && hw == 8)),
                 "gims_ch_frn_block: bad arguments (32x32x32, 16x16x64 or 8x8x128 activations)");
  GIMS_CHECK_ARG(!y_split || (ld_split >= 2 * (int64_t)c && (ld_split % 4) == 0), "gims_ch_frn_block: split output pitch >= 2c");
  ChGateW G = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (gate_w) {
    for (int i = 0; i < 6; ++i) GIMS_CHECK_ARG(gate_w[i] != nullptr, "gims_ch_frn_block: gate weight pointer %d is null", i);
    G = ChGateW{gate_w[0], gate_w[1], gate_w[2], gate_w[3], gate_w[4], gate_w[5]};
  }
  constexpr int FRN_NT = 1024;
  const size_t lds = ((size_t)hw * hw * c + (FRN_NT / c) * (size_t)c + c + 2 * (size_t)hw * c + 16 * (size_t)hw) * sizeof(float);
  GIMS_LDS_ATTR((const void*)ch_frn_block_kernel<32, 32, FRN_NT>, 160 * 1024);
  GIMS_LDS_ATTR((const void*)ch_frn_block_kernel<64, 16, FRN_NT>, 160 * 1024);
  GIMS_LDS_ATTR((const void*)ch_frn_block_kernel<128, 8, FRN_NT>, 160 * 1024);
  const dim3 grid((unsigned)patches);
  hipStream_t st = (hipStream_t)stream;
  if (c == 32) hipLaunchKernelGGL((ch_frn_block_kernel<32, 32, FRN_NT>), grid, dim3(FRN_NT), lds, st, x, frn_weight, frn_bias, eps, G, tau, y, y_split, ld_split);
  else if (c == 64) hipLaunchKernelGGL((ch_frn_block_kernel<64, 16, FRN_NT>), grid, dim3(FRN_NT), lds, st, x, frn_weight, frn_bias, eps, G, tau, y, y_split, ld_split);
  else hipLaunchKernelGGL((ch_frn_block_kernel<128, 8, FRN_NT>), grid, dim3(FRN_NT), lds, st, x, frn_weight, frn_bias, eps, G, tau, y, y_split, ld_split);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

constexpr int CH_PP5 = 1, CH_PP6 = 2;      // patches per workgroup of the 16x16x64 -> 8x8x128 layer (two measured slower: 357 -> 374 us) and of the 8x8x128 layer (537 -> 505 us per 8192 patches)
template <int CIN, int COUT, int HIN, int STRIDE, bool FIRST = false, int PP = 1>
static int conv_block_launch(const uint16_t* x, int64_t ldx, int64_t patches, const uint16_t* w, const float* bias, const float* fw, const float* fb, float eps,
                             gims::ChGateW G, const float* tau, float* y, uint16_t* ysp, int64_t ldsp, hipStream_t st, gims::ChFirst first = {nullptr, nullptr, nullptr, 0.f, nullptr, 0, 0}) {
  using namespace gims;
  using Geo = ConvGeom<CIN, COUT, HIN, STRIDE, PP>;
  GIMS_LDS_ATTR((const void*)ch_conv_block_kernel<CIN, COUT, HIN, STRIDE, FIRST, PP>, Geo::LDS_BYTES + Geo::GATE_BYTES);
  static const bool prof_on = getenv("GIMS_CH_PROF") != nullptr;      // diagnostics: cycle stamps of one workgroup per launch (synchronous)
  unsigned long long* dprof = nullptr;
  if (prof_on) {
    dprof = (unsigned long long*)device_once("ch_conv_prof", 8 * sizeof(unsigned long long), nullptr);
    GIMS_CHECK_ARG(dprof, "gims_ch_conv_block: no profile buffer");
    first.prof = dprof;
  }
  static const int stagger = getenv("GIMS_CH_STAGGER") ? atoi(getenv("GIMS_CH_STAGGER")) : 8000;
  first.stagger = stagger;
  // (FRN_FLOATS * 4 <= LDS_BYTES by construction; the staged gate weights sit behind the FRN arrays)
  const int lds_bytes = G.w1 ? PP * Geo::FRN_FLOATS * 4 + Geo::GATE_BYTES : Geo::LDS_BYTES;
  const int lds_launch = lds_bytes > Geo::LDS_BYTES ? lds_bytes : Geo::LDS_BYTES;
  first.first_wave = 256 * (int)((160 * 1024) / lds_launch);
  hipLaunchKernelGGL((ch_conv_block_kernel<CIN, COUT, HIN, STRIDE, FIRST, PP>), dim3((unsigned)cdiv(patches, PP)), dim3(512), lds_launch, st, x, ldx, w, bias, fw, fb,
                     eps, G, tau, y, ysp, ldsp, first, (int)patches);
  GIMS_LAUNCH_CHECK();
  if (prof_on) {
    unsigned long long h[8];
    GIMS_HIP(hipStreamSynchronize(st));
    GIMS_HIP(hipMemcpy(h, dprof, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "ch_conv_block<%d,%d,%d,%d,%d> x %lld: load %llu  mfma %llu  acc->lds %llu  frn-stat %llu  pools+gates %llu  apply+store %llu  (cycles of one workgroup)\n",
            CIN, COUT, HIN, STRIDE, (int)FIRST, (long long)patches, h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5]);
  }
  return GIMS_OK;
}

// the two-workgroups-per-CU form of the 32 x 32 layers (ch_conv_block_half_kernel); GIMS_CH_HALF=0: the one-workgroup kernel (the cross-check)
template <int CIN, int COUT, int STRIDE>
static int conv_block_half_launch(const uint16_t* x, int64_t ldx, int64_t patches, const uint16_t* w, const float* bias, const float* fw, const float* fb, float eps,
                                  gims::ChGateW G, const float* tau, float* y, uint16_t* ysp, int64_t ldsp, hipStream_t st) {
  using namespace gims;
  using Geo = HalfGeom<CIN, COUT, STRIDE>;
  GIMS_LDS_ATTR((const void*)ch_conv_block_half_kernel<CIN, COUT, STRIDE>, Geo::LDS_BYTES);
  static const int stagger = getenv("GIMS_CH_STAGGER") ? atoi(getenv("GIMS_CH_STAGGER")) : 8000;
  static const bool prof_on = getenv("GIMS_CH_PROF") != nullptr;      // diagnostics: cycle stamps of one workgroup per launch (synchronous)
  unsigned long long* dprof = nullptr;
  if (prof_on) {
    dprof = (unsigned long long*)device_once("ch_conv_half_prof", 8 * sizeof(unsigned long long), nullptr);
    GIMS_CHECK_ARG(dprof, "gims_ch_conv_block: no profile buffer");
  }
  // GIMS_CH_HALF_LDS=<bytes>: pad the dynamic LDS (diagnostics: > 80 KB leaves ONE workgroup per CU -- what the co-residency is worth)
  static const int pad = getenv("GIMS_CH_HALF_LDS") ? atoi(getenv("GIMS_CH_HALF_LDS")) : 0;
  const int lds = pad > Geo::LDS_BYTES ? pad : Geo::LDS_BYTES;
  if (pad) GIMS_LDS_ATTR((const void*)ch_conv_block_half_kernel<CIN, COUT, STRIDE>, lds);
  hipLaunchKernelGGL((ch_conv_block_half_kernel<CIN, COUT, STRIDE>), dim3((unsigned)patches), dim3(512), lds, st, x, ldx, w, bias, fw, fb, eps, G, tau, y,
                     ysp, ldsp, stagger, 2 * device_cus(), dprof);
  GIMS_LAUNCH_CHECK();
  if (prof_on) {
    unsigned long long h[8];
    GIMS_HIP(hipStreamSynchronize(st));
    GIMS_HIP(hipMemcpy(h, dprof, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "ch_conv_block_half<%d,%d,%d> x %lld: load0 %llu  mfma0 %llu  load1 %llu  mfma1 %llu  dumps+stats+pools %llu  gates %llu  dumps+apply+store %llu  total %llu (cycles of one workgroup)\n",
            CIN, COUT, STRIDE, (long long)patches, h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6], h[7] - h[0]);
  }
  return GIMS_OK;
}
static bool ch_half() {
  const char* e = getenv("GIMS_CH_HALF");         // read per call: the tests compare the two forms
  return !e || atoi(e) != 0;
}

extern "C" int gims_ch_conv_block_first(const float* patches, int64_t n, const float* frn0_weight, const float* frn0_bias, float eps0, const float* tau0,
                                        const uint16_t* w_packed, const float* bias, const float* frn_weight, const float* frn_bias, float eps,
                                        const float* const* gate_w, const float* tau, float* y, uint16_t* y_split, int64_t ld_split, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(patches && frn0_weight && frn0_bias && tau0 && w_packed && bias && frn_weight && frn_bias && tau && (y || y_split) && n > 0 && eps >= 0.f &&
                     eps0 >= 0.f, "gims_ch_conv_block_first: null pointer / empty batch");
  GIMS_CHECK_ARG(!y_split || (ld_split >= 64 && (ld_split % 4) == 0), "gims_ch_conv_block_first: split output pitch >= 64");
  ChGateW G = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (gate_w) {
    for (int i = 0; i < 6; ++i) GIMS_CHECK_ARG(gate_w[i] != nullptr, "gims_ch_conv_block_first: gate weight pointer %d is null", i);
    G = ChGateW{gate_w[0], gate_w[1], gate_w[2], gate_w[3], gate_w[4], gate_w[5]};
  }
  return conv_block_launch<16, 32, 32, 1, true>((const uint16_t*)patches, 0, n, w_packed, bias, frn_weight, frn_bias, eps, G, tau, y, y_split, ld_split,
                                                 (hipStream_t)stream, ChFirst{frn0_weight, frn0_bias, tau0, eps0, nullptr});
}

extern "C" int gims_ch_conv_block(const uint16_t* x_split, int64_t ldx, int64_t patches, int32_t hin, int32_t cin, int32_t cout, int32_t stride,
                                  const uint16_t* w_packed, const float* bias, const float* frn_weight, const float* frn_bias, float eps,
                                  const float* const* gate_w, const float* tau, float* y, uint16_t* y_split, int64_t ld_split, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(x_split && w_packed && bias && frn_weight && frn_bias && tau && (y || y_split) && patches > 0 && eps >= 0.f,
                 "gims_ch_conv_block: null pointer / empty batch");
  GIMS_CHECK_ARG(ldx >= 2 * (int64_t)cin && (ldx % 8) == 0 && (((uintptr_t)x_split) & 15) == 0, "gims_ch_conv_block: input pitch >= 2 cin, 16-byte aligned rows");
  GIMS_CHECK_ARG(!y_split || (ld_split >= 2 * (int64_t)cout && (ld_split % 4) == 0), "gims_ch_conv_block: split output pitch >= 2 cout");
  ChGateW G = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (gate_w) {
    for (int i = 0; i < 6; ++i) GIMS_CHECK_ARG(gate_w[i] != nullptr, "gims_ch_conv_block: gate weight pointer %d is null", i);
    G = ChGateW{gate_w[0], gate_w[1], gate_w[2], gate_w[3], gate_w[4], gate_w[5]};
  }
  hipStream_t st = (hipStream_t)stream;
  const int key = hin * 1000000 + cin * 10000 + cout * 10 + stride;
  switch (key) {
    case 32 * 1000000 + 32 * 10000 + 32 * 10 + 1:
      if (ch_half()) return conv_block_half_launch<32, 32, 1>(x_split, ldx, patches, w_packed, bias, frn_weight, frn_bias, eps, G, tau, y, y_split, ld_split, st);
      return conv_block_launch<32, 32, 32, 1>(x_split, ldx, patches, w_packed, bias, frn_weight, frn_bias, eps, G, tau, y, y_split, ld_split, st);
    case 32 * 1000000 + 32 * 10000 + 64 * 10 + 2:
      if (ch_half()) return conv_block_half_launch<32, 64, 2>(x_split, ldx, patches, w_packed, bias, frn_weight, frn_bias, eps, G, tau, y, y_split, ld_split, st);
      return conv_block_launch<32, 64, 32, 2>(x_split, ldx, patches, w_packed, bias, frn_weight, frn_bias, eps, G, tau, y, y_split, ld_split, st);
    case 16 * 1000000 + 64 * 10000 + 64 * 10 + 1: return conv_block_launch<64, 64, 16, 1>(x_split, ldx, patches, w_packed, bias, frn_weight, frn_bias, eps, G, tau, y, y_split, ld_split, st);
    case 16 * 1000000 + 64 * 10000 + 128 * 10 + 2: return conv_block_launch<64, 128, 16, 2, false, CH_PP5>(x_split, ldx, patches, w_packed, bias, frn_weight, frn_bias, eps, G, tau, y, y_split, ld_split, st);
    case 8 * 1000000 + 128 * 10000 + 128 * 10 + 1: return conv_block_launch<128, 128, 8, 1, false, CH_PP6>(x_split, ldx, patches, w_packed, bias, frn_weight, frn_bias, eps, G, tau, y, y_split, ld_split, st);
    default: break;
  }
  GIMS_CHECK_ARG(false, "gims_ch_conv_block: unsupported geometry hin=%d cin=%d cout=%d stride=%d (the five 3x3 layers of CAR-HyNet after the first)", hin, cin, cout, stride);
}

extern "C" int gims_ch_sandglass(const float* x, int64_t patches, int32_t hw, int32_t c, const float* const* w /* 14 device pointers, ChSandglassW order */,
                                 uint16_t* out_split, int64_t ld_split, void* stream) {
  GIMS_CHECK_ARG(x && w && out_split && patches > 0 && ((c == 32 && hw == 32) || (c == 64 && hw == 16)) && ld_split >= 2 * (int64_t)c && (ld_split % 4) == 0,
                 "gims_ch_sandglass: bad arguments (32x32x32 or 16x16x64 activations)");
  ChSandglassW W;
  const float** dst = (const float**)&W;
  for (int i = 0; i < 14; ++i) { GIMS_CHECK_ARG(w[i] != nullptr, "gims_ch_sandglass: weight pointer %d is null", i); dst[i] = w[i]; }
  constexpr int SG_NT = 1024;
  const size_t lds = ((size_t)hw * hw * c + 2 * (size_t)hw * c + 16 * (size_t)hw + (size_t)c * (9 + 1 + 9 + 1 + 16 + 20 + 1 + 8 + 8 + 8 + 1 + 1) + 16 * 4 + 16 + 8) * sizeof(float);
  GIMS_LDS_ATTR((const void*)ch_sandglass_kernel<32, 32, SG_NT>, 160 * 1024);
  GIMS_LDS_ATTR((const void*)ch_sandglass_kernel<64, 16, SG_NT>, 160 * 1024);
  static const bool prof_on = getenv("GIMS_CH_PROF") != nullptr;
  unsigned long long* dprof = nullptr;
  W.prof = nullptr;
  if (prof_on) {
    dprof = (unsigned long long*)device_once("ch_sandglass_prof", 8 * sizeof(unsigned long long), nullptr);
    GIMS_CHECK_ARG(dprof, "gims_ch_sandglass: no profile buffer");
    W.prof = dprof;
  }
  static const int stagger = getenv("GIMS_CH_STAGGER") ? atoi(getenv("GIMS_CH_STAGGER")) : 8000;
  W.stagger = stagger;
  W.first_wave = 256 * (int)((160 * 1024) / lds);
  // 32x32x32: one 1024-thread workgroup per CU, persistent over the patches (806 -> 691 us per 8192 patches); 16x16x64: one workgroup per patch
  static const int persist = getenv("GIMS_CH_PERSIST") ? atoi(getenv("GIMS_CH_PERSIST")) : 1;
  const unsigned grid = (unsigned)(c == 32 && persist && patches > 256 ? 256 : patches);
  if (c == 32) hipLaunchKernelGGL((ch_sandglass_kernel<32, 32, SG_NT>), dim3(grid), dim3(SG_NT), lds, (hipStream_t)stream, x, W, out_split, ld_split, (int)patches);
  else hipLaunchKernelGGL((ch_sandglass_kernel<64, 16, SG_NT>), dim3(grid), dim3(SG_NT), lds, (hipStream_t)stream, x, W, out_split, ld_split, (int)patches);
  GIMS_LAUNCH_CHECK();
  if (prof_on) {
    unsigned long long h[8];
    GIMS_HIP(hipStreamSynchronize((hipStream_t)stream));
    GIMS_HIP(hipMemcpy(h, dprof, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "ch_sandglass<%d,%d> x %lld: weights->lds %llu  patch load %llu  A dw3x3 %llu  A2 pools %llu  B gates %llu  C pointwise %llu  D dw3x3+store %llu  (cycles of one workgroup)\n",
            c, hw, (long long)patches, h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6]);
  }
  return GIMS_OK;
}

extern "C" int gims_ch_l2norm(const float* x, int64_t rows, int32_t c, float eps, float* y, void* stream) {
  GIMS_CHECK_ARG(x && y && rows > 0 && c > 0 && c <= 256, "gims_ch_l2norm: bad arguments (c <= 256)");
  hipLaunchKernelGGL(ch_l2norm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, rows, c, eps, y);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_ch_relu6(float* x, int64_t total, void* stream) {
  GIMS_CHECK_ARG(x && total > 0 && (total % 4) == 0, "gims_ch_relu6: bad arguments (total %% 4 == 0)");
  hipLaunchKernelGGL(ch_relu6_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, total);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

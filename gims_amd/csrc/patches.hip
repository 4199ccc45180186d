// Patch extraction on the device: the stage of the reference's front end between keypoint detection
// and the CAR-HyNet descriptor network (utils/common.py:882-884) --
//     pyramid = buildGaussianPyramid(img, 6, graydesc=False)                 utils/library.py:234-271
//     pts     = ComputePatches(k, pyramid, radius_size=64)                   utils/library.py:84-110
//     pts     = [cv2.resize(p, (32, 32), INTER_AREA) for p in pts] / 255.0
// which the reference runs on one CPU core through OpenCV (README.md:151,155: 3.2-3.9 s per image at 15 k keypoints).
//
// The arithmetic is OpenCV's uint8 fixed-point arithmetic, restated (PARITY UNPINNED against the library itself, which is
// absent here; bit-identical to the NumPy restatement of the same published algorithms that the tests hold):
//   * 2x upsampling, INTER_LINEAR_EXACT: weights 1/4, 3/4 per direction, one rounding (half up) of the 16ths;
//   * Gaussian blur on uint8: Q8.8 kernel with error diffusion (host), row pass Q8.8, column pass Q16.16, one rounding,
//     BORDER_REFLECT_101;
//   * octave step: nearest-neighbour decimation by 2;
//   * warpAffine(INTER_CUBIC, BORDER_CONSTANT): inverse map in double, coordinates in 1/32 pixel (10-bit fixed point),
//     32 x 32 table of 4 x 4 15-bit bicubic weights (A = -0.75, float32 construction, sums forced to 2^15);
//   * INTER_AREA 64 -> 32 on float32 = the 2 x 2 mean, then / 255.
// Images are uint8 HWC; every level of the pyramid lives in one buffer (gims_pyramid_layout gives the offsets).
// One workgroup per keypoint warps its 64 x 64 x 3 patch into LDS and writes the 32 x 32 x 3 float32 patch -- the NHWC input
// layout of gims_amd.carhynet.CARHyNet -- so the patches never exist on the host.
#include "common.h"

#include <math.h>
#include <vector>

namespace gims {

constexpr int PYR_LAYERS = 6;          // nOctaveLayers + 3 (library.py:238, 261)
struct BlurKernel { int n; int q[33]; };

__global__ void up2x_kernel(const uint8_t* __restrict__ src, int h, int w, int c, uint8_t* __restrict__ dst) {
  const int64_t total = (int64_t)4 * h * w * c;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    const int64_t p = i / c;
    const int dx = (int)(p % (2 * w)), dy = (int)(p / (2 * w));
    const int kx = dx >> 1, ky = dy >> 1;
    int x0 = (dx & 1) ? kx : kx - 1, y0 = (dy & 1) ? ky : ky - 1;
    const int wx1 = (dx & 1) ? 1 : 3, wy1 = (dy & 1) ? 1 : 3;        // weight of the second tap, in quarters
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : x0; y0 = y0 < 0 ? 0 : y0;
    x1 = x1 > w - 1 ? w - 1 : x1; y1 = y1 > h - 1 ? h - 1 : y1;
    const int a = src[((int64_t)y0 * w + x0) * c + ch], b = src[((int64_t)y0 * w + x1) * c + ch];
    const int d = src[((int64_t)y1 * w + x0) * c + ch], e = src[((int64_t)y1 * w + x1) * c + ch];
    const int s = (4 - wy1) * ((4 - wx1) * a + wx1 * b) + wy1 * ((4 - wx1) * d + wx1 * e);
    dst[i] = (uint8_t)((s + 8) >> 4);
  }
}

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  const int p = 2 * (n - 1);
  i %= p;
  if (i < 0) i += p;
  return i >= n ? p - i : i;
}

__global__ void blur_row_kernel(const uint8_t* __restrict__ src, int h, int w, int c, BlurKernel k, uint16_t* __restrict__ dst) {
  const int64_t total = (int64_t)h * w * c;
  const int r = k.n >> 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    const int64_t p = i / c;
    const int x = (int)(p % w);
    const int64_t row = (p / w) * w;
    int s = 0;
    for (int t = 0; t < k.n; ++t) s += k.q[t] * (int)src[(row + reflect101(x + t - r, w)) * c + ch];
    dst[i] = (uint16_t)s;            // Q8.8, <= 255 * 256
  }
}

__global__ void blur_col_kernel(const uint16_t* __restrict__ src, int h, int w, int c, BlurKernel k, uint8_t* __restrict__ dst) {
  const int64_t total = (int64_t)h * w * c;
  const int r = k.n >> 1;
  const int64_t pitch = (int64_t)w * c;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / pitch);
    const int64_t xc = i % pitch;
    int64_t s = 0;
    for (int t = 0; t < k.n; ++t) s += (int64_t)k.q[t] * src[(int64_t)reflect101(y + t - r, h) * pitch + xc];
    s = (s + (1 << 15)) >> 16;
    dst[i] = (uint8_t)(s > 255 ? 255 : s);
  }
}

__global__ void half_kernel(const uint8_t* __restrict__ src, int h, int w, int c, int nh, int nw, uint8_t* __restrict__ dst) {
  const int64_t total = (int64_t)nh * nw * c;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    const int64_t p = i / c;
    int x = 2 * (int)(p % nw), y = 2 * (int)(p / nw);
    x = x > w - 1 ? w - 1 : x; y = y > h - 1 ? h - 1 : y;
    dst[i] = src[((int64_t)y * w + x) * c + ch];
  }
}

// ---- warpAffine(INTER_CUBIC, BORDER_CONSTANT) + INTER_AREA halving + / 255, one workgroup per keypoint
constexpr int PATCH_DIM = 64;           // int32(2 * ((64 - 1) / 2) + 1)   (library.py:91-93 with radius_size = 64)

// Host arithmetic of ComputePatches for one keypoint (library.py:96-106): the 2x3 map A it hands to cv2.warpAffine, float64
// except where the reference itself drops to float32, and the pyramid level index (octave - firstOctave) * 6 + layer
// (may lie outside the pyramid).  Returns the layer.  Pinned bit for bit by tests/golden/patch_affine_*.npz through
// gims_patch_affine.
__device__ __forceinline__ int keypoint_affine(const float* __restrict__ kp4, const int32_t* __restrict__ kp_oct, int kp, double* A, int* level) {
#pragma clang fp contract(off)
  const double x = kp4[4 * kp], y = kp4[4 * kp + 1], size = kp4[4 * kp + 2], angle_in = kp4[4 * kp + 3];
  const int packed = kp_oct[kp];
  int octave = packed & 0xFF;
  const int layer = (packed >> 8) & 0xFF;
  if (octave >= 128) octave |= -128;
  const double scale = octave >= 0 ? 1.0 / (double)(1 << octave) : (double)(1 << -octave);
  const double step = size * scale * 0.5;
  const double px = x * scale, py = y * scale;
  double ang = 360.0 - angle_in;
  if (fabs(ang - 360.0) < 1.19209e-07) ang = 0.0;
  const double phi = ang * (3.14159265358979323846 / 180.0);
  const double s = sin(phi), c = cos(phi);
  const float stepf = (float)step;
  A[0] = (double)((float)c / stepf); A[1] = (double)((float)(-s) / stepf); A[3] = (double)((float)s / stepf); A[4] = (double)((float)c / stepf);
  const double r = (64 - 1) / 2.0;
  A[2] = r - (A[0] * px + A[1] * py);
  A[5] = r - (A[3] * px + A[4] * py);
  *level = (octave + 1) * PYR_LAYERS + layer;
  return layer;
}

__global__ void patch_affine_kernel(const float* __restrict__ kp4, const int32_t* __restrict__ kp_oct, int n_kp, double* __restrict__ A_out,
                                    int32_t* __restrict__ level_out) {
  const int kp = blockIdx.x * blockDim.x + threadIdx.x;
  if (kp >= n_kp) return;
  double A[6];
  int level;
  keypoint_affine(kp4, kp_oct, kp, A, &level);
  for (int i = 0; i < 6; ++i) A_out[6 * kp + i] = A[i];
  level_out[kp] = level;
}
__global__ __launch_bounds__(256) void patch_kernel(const uint8_t* __restrict__ pyr, const gims_pyr_level* __restrict__ levels, int n_levels,
                                                    const float* __restrict__ kp4, const int32_t* __restrict__ kp_oct, int n_kp,
                                                    const int16_t* __restrict__ wtab, float* __restrict__ out, int32_t* __restrict__ bad) {
#pragma clang fp contract(off)
  __shared__ uint8_t patch[PATCH_DIM * PATCH_DIM * 3];
  __shared__ double minv[6];
  __shared__ int lvl;
  const int kp = blockIdx.x;
  if (kp >= n_kp) return;
  if (threadIdx.x == 0) {
    double A[6];
    int l;
    const int layer = keypoint_affine(kp4, kp_oct, kp, A, &l);
    const double a00 = A[0], a01 = A[1], m2 = A[2], a10 = A[3], a11 = A[4], m5 = A[5];
    // cv::warpAffine inverts the map in double (imgwarp.cpp)
    double D = a00 * a11 - a01 * a10;
    D = D != 0 ? 1.0 / D : 0.0;
    const double i0 = a11 * D, i4 = a00 * D, i1 = a01 * -D, i3 = a10 * -D;
    minv[0] = i0; minv[1] = i1; minv[3] = i3; minv[4] = i4;
    minv[2] = -i0 * m2 - i1 * m5;
    minv[5] = -i3 * m2 - i4 * m5;
    lvl = (l >= 0 && l < n_levels && layer < PYR_LAYERS) ? l : -1;
    if (lvl < 0) atomicAdd(bad, 1);
  }
  __syncthreads();
  const int l = lvl;
  if (l < 0) {            // keypoint outside the pyramid: zero patch, counted in *bad (the reference would raise IndexError)
    for (int i = threadIdx.x; i < 32 * 32 * 3; i += 256) out[(int64_t)kp * 3072 + i] = 0.f;
    return;
  }
  const gims_pyr_level L = levels[l];
  const uint8_t* img = pyr + L.offset;
  const int h = L.h, w = L.w;
  for (int p = threadIdx.x; p < PATCH_DIM * PATCH_DIM; p += 256) {
    const int dx = p & 63, dy = p >> 6;
    const long long adx = llrint(minv[0] * (double)dx * 1024.0), bdx = llrint(minv[3] * (double)dx * 1024.0);
    const long long X0 = llrint((minv[1] * (double)dy + minv[2]) * 1024.0) + 16, Y0 = llrint((minv[4] * (double)dy + minv[5]) * 1024.0) + 16;
    const long long X = (X0 + adx) >> 5, Y = (Y0 + bdx) >> 5;
    const long long sx = (X >> 5) - 1, sy = (Y >> 5) - 1;
    const int16_t* wt = wtab + (((int)(Y & 31)) * 32 + (int)(X & 31)) * 16;
    int acc0 = 0, acc1 = 0, acc2 = 0;
    if (sx > -4 && sx < w && sy > -4 && sy < h) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long yy = sy + i;
        if (yy < 0 || yy >= h) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long long xx = sx + j;
          if (xx < 0 || xx >= w) continue;
          const uint8_t* px = img + ((int64_t)yy * w + xx) * 3;
          const int wv = wt[4 * i + j];
          acc0 += wv * px[0]; acc1 += wv * px[1]; acc2 += wv * px[2];
        }
      }
    }
    auto cast = [](int v) { v = (v + (1 << 14)) >> 15; return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
    patch[3 * p] = cast(acc0); patch[3 * p + 1] = cast(acc1); patch[3 * p + 2] = cast(acc2);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * 32 * 3; i += 256) {
    const int ch = i % 3, q = i / 3, ox = q & 31, oy = q >> 5;
    const uint8_t* a = patch + ((2 * oy) * PATCH_DIM + 2 * ox) * 3 + ch;
    const float s = ((float)a[0] + (float)a[3] + (float)a[PATCH_DIM * 3] + (float)a[PATCH_DIM * 3 + 3]) * 0.25f;
    out[(int64_t)kp * 3072 + i] = s / 255.0f;
  }
}

// ---- host side
static BlurKernel make_blur_kernel(double sigma) {
  BlurKernel k{};
  int n = (int)nearbyint(sigma * 6 + 1) | 1;
  if (n > 33) n = 33;
  std::vector<double> v(n);
  double sum = 0;
  for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; v[i] = exp(-(x * x) / (2.0 * sigma * sigma)); sum += v[i]; }
  double err = 0;
  int acc = 0;
  for (int i = 0; i < n / 2; ++i) {
    const double adj = v[i] / sum * 256.0 + err;
    const int q = (int)nearbyint(adj);
    err = adj - q;
    k.q[i] = k.q[n - 1 - i] = q;
    acc += q;
  }
  k.q[n / 2] = 256 - 2 * acc;
  k.n = n;
  return k;
}

static void layer_sigmas(double* sig) {            // library.py:252-257, with its float32 / double mix
  sig[0] = 1.6;
  const float kf = (float)pow(2.0, 1.0 / (double)3.0f);
  for (int i = 1; i < PYR_LAYERS; ++i) {
    // pow(k, np.float32(i - 1)): numpy float32 ** float32 -> float32, times the python float sigma -> float64
    const double sig_prev = (double)powf(kf, (float)(i - 1)) * 1.6;
    const double sig_total = sig_prev * (double)kf;
    sig[i] = sqrt(sig_total * sig_total - sig_prev * sig_prev);
  }
}

static int n_octaves(int h2, int w2) {             // library.py:248-250 on the DOUBLED image
  const int mn = h2 < w2 ? h2 : w2;
  const double v = (double)logf((float)mn) / log(2.0) - 2.0;
  return (int)nearbyint(v) + 1;
}

static std::vector<int16_t> cubic_table_host() {
#pragma clang fp contract(off)
  std::vector<float> t1(32 * 4);
  for (int i = 0; i < 32; ++i) {
    const float x = (float)i * (1.0f / 32), A = -0.75f;
    float* c = &t1[i * 4];
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
  }
  std::vector<int16_t> tab(32 * 32 * 16);
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      int16_t* w = &tab[(i * 32 + j) * 16];
      int isum = 0;
      for (int k1 = 0; k1 < 4; ++k1)
        for (int k2 = 0; k2 < 4; ++k2) {
          const float v = t1[i * 4 + k1] * t1[j * 4 + k2];
          long q = lrintf(v * 32768.0f);
          q = q < -32768 ? -32768 : (q > 32767 ? 32767 : q);
          w[k1 * 4 + k2] = (int16_t)q;
          isum += (int)q;
        }
      if (isum != 32768) {
        const int diff = isum - 32768;
        int Mk = 2 * 4 + 2, mk = 2 * 4 + 2;
        for (int k1 = 2; k1 < 4; ++k1)
          for (int k2 = 2; k2 < 4; ++k2) {
            if (w[k1 * 4 + k2] < w[mk]) mk = k1 * 4 + k2;
            else if (w[k1 * 4 + k2] > w[Mk]) Mk = k1 * 4 + k2;
          }
        if (diff < 0) w[Mk] = (int16_t)(w[Mk] - diff); else w[mk] = (int16_t)(w[mk] - diff);
      }
    }
  return tab;
}
// one device copy per device, created under a lock (the table is a constant of the algorithm, 32 KB; the host copy is built once per process)
static const int16_t* cubic_table() {
  static const std::vector<int16_t> tab = cubic_table_host();
  return (const int16_t*)device_once("patch_cubic_table", tab.size() * sizeof(int16_t), tab.data());
}

static int pyramid_levels(int h, int w, int c, std::vector<gims_pyr_level>& lv, size_t& bytes) {
  const int no = n_octaves(2 * h, 2 * w);
  if (no < 1) return 0;
  lv.clear();
  bytes = 0;
  int ch = 2 * h, cw = 2 * w;
  for (int o = 0; o < no; ++o) {
    if (o > 0) { ch = (int)nearbyint(ch * 0.5); cw = (int)nearbyint(cw * 0.5); }
    if (ch < 1 || cw < 1) break;
    for (int i = 0; i < PYR_LAYERS; ++i) {
      gims_pyr_level L; L.offset = (int64_t)bytes; L.h = ch; L.w = cw;
      lv.push_back(L);
      bytes += (((size_t)ch * cw * c) + 255) & ~(size_t)255;
    }
  }
  return (int)lv.size();
}

}  // namespace gims

extern "C" int gims_pyramid_layout(int32_t h, int32_t w, int32_t c, gims_pyr_level* h_levels, int32_t cap, int32_t* n_levels, size_t* pyr_bytes,
                                   size_t* scratch_bytes) {
  using namespace gims;
  GIMS_CHECK_ARG(h > 0 && w > 0 && c > 0 && n_levels && pyr_bytes && scratch_bytes, "gims_pyramid_layout: bad arguments");
  std::vector<gims_pyr_level> lv;
  size_t bytes = 0;
  const int n = pyramid_levels(h, w, c, lv, bytes);
  GIMS_CHECK_ARG(n > 0, "gims_pyramid_layout: image too small for a pyramid");
  *n_levels = n;
  *pyr_bytes = bytes;
  *scratch_bytes = (size_t)4 * h * w * c * sizeof(uint16_t);       // Q8.8 row-pass image of the largest level
  if (h_levels) {
    GIMS_CHECK_ARG(cap >= n, "gims_pyramid_layout: level table too small (%d < %d)", cap, n);
    for (int i = 0; i < n; ++i) h_levels[i] = lv[i];
  }
  return GIMS_OK;
}

extern "C" int gims_pyramid_build(const uint8_t* img, int32_t h, int32_t w, int32_t c, uint8_t* pyr, void* scratch, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(img && pyr && scratch && h > 0 && w > 0 && c > 0, "gims_pyramid_build: bad arguments");
  std::vector<gims_pyr_level> lv;
  size_t bytes = 0;
  const int n = pyramid_levels(h, w, c, lv, bytes);
  GIMS_CHECK_ARG(n > 0, "gims_pyramid_build: image too small for a pyramid");
  hipStream_t s = (hipStream_t)stream;
  double sig[PYR_LAYERS];
  layer_sigmas(sig);
  BlurKernel bk[PYR_LAYERS];
  for (int i = 1; i < PYR_LAYERS; ++i) bk[i] = make_blur_kernel(sig[i]);
  auto grid = [](int64_t total) { int64_t g = (total + 255) / 256; return dim3((unsigned)(g > 65535 * 4 ? 65535 * 4 : (g < 1 ? 1 : g))); };
  hipLaunchKernelGGL(up2x_kernel, grid((int64_t)4 * h * w * c), dim3(256), 0, s, img, h, w, c, pyr + lv[0].offset);
  for (int l = 1; l < n; ++l) {
    const int i = l % PYR_LAYERS;
    const gims_pyr_level& L = lv[l];
    const int64_t total = (int64_t)L.h * L.w * c;
    if (i == 0) {
      const gims_pyr_level& S = lv[l - PYR_LAYERS + 3];                 // pyr[(o - 1) * 6 + nOctaveLayers]
      hipLaunchKernelGGL(half_kernel, grid(total), dim3(256), 0, s, pyr + S.offset, S.h, S.w, c, L.h, L.w, pyr + L.offset);
    } else {
      hipLaunchKernelGGL(blur_row_kernel, grid(total), dim3(256), 0, s, pyr + lv[l - 1].offset, L.h, L.w, c, bk[i], (uint16_t*)scratch);
      hipLaunchKernelGGL(blur_col_kernel, grid(total), dim3(256), 0, s, (const uint16_t*)scratch, L.h, L.w, c, bk[i], pyr + L.offset);
    }
  }
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_patch_affine(const float* kp4, const int32_t* kp_octave, int32_t n_kp, double* A_out, int32_t* level_out, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(n_kp >= 0 && (n_kp == 0 || (kp4 && kp_octave && A_out && level_out)), "gims_patch_affine: bad arguments");
  if (n_kp == 0) return GIMS_OK;
  hipLaunchKernelGGL(patch_affine_kernel, dim3(cdiv(n_kp, 64)), dim3(64), 0, (hipStream_t)stream, kp4, kp_octave, n_kp, A_out, level_out);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_patch_extract(const uint8_t* pyr, const gims_pyr_level* dev_levels, int32_t n_levels, const float* kp4, const int32_t* kp_octave,
                                  int32_t n_kp, float* out, int32_t* bad_count, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(pyr && dev_levels && n_levels > 0 && n_kp >= 0 && bad_count && (n_kp == 0 || (kp4 && kp_octave && out)), "gims_patch_extract: bad arguments");
  const int16_t* wt = cubic_table();
  GIMS_CHECK_ARG(wt, "gims_patch_extract: could not create the bicubic weight table");
  GIMS_HIP(hipMemsetAsync(bad_count, 0, sizeof(int32_t), (hipStream_t)stream));
  if (n_kp > 0)
    hipLaunchKernelGGL(patch_kernel, dim3(n_kp), dim3(256), 0, (hipStream_t)stream, pyr, dev_levels, n_levels, kp4, kp_octave, n_kp, wt, out, bad_count);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

// Small HBM/latency-bound kernels of the matcher path: keypoint-encoder front end, GraphSAGE mean
// aggregation, row gather.  One wave per row, float4 lanes, no LDS.
#include "common.h"

#include <stdarg.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <utility>

namespace gims {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  return dev;
}

int device_cus() {
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, current_device()) != hipSuccess || n < 8) n = 8;
  return n;
}

static std::mutex g_once_mu;

int lds_attr(const void* fn, int bytes) {
  static std::map<std::pair<const void*, int>, int> done;
  const int dev = current_device();
  std::lock_guard<std::mutex> lock(g_once_mu);
  int& have = done[std::make_pair(fn, dev)];
  if (have >= bytes) return GIMS_OK;
  GIMS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  have = bytes;
  return GIMS_OK;
}

void* device_once(const char* key, size_t bytes, const void* init) {
  static std::map<std::pair<std::string, int>, void*> bufs;
  const int dev = current_device();
  std::lock_guard<std::mutex> lock(g_once_mu);
  void*& p = bufs[std::make_pair(std::string(key), dev)];
  if (p) return p;
  void* d = nullptr;
  if (hipMalloc(&d, bytes) != hipSuccess) return nullptr;
  if ((init ? hipMemcpy(d, init, bytes, hipMemcpyHostToDevice) : hipMemset(d, 0, bytes)) != hipSuccess) { (void)hipFree(d); return nullptr; }
  p = d;
  return p;
}

void* pinned_once(const char* key, size_t bytes) {
  static std::map<std::pair<std::string, int>, void*> bufs;
  const int dev = current_device();
  std::lock_guard<std::mutex> lock(g_once_mu);
  void*& p = bufs[std::make_pair(std::string(key), dev)];
  if (p) return p;
  void* h = nullptr;
  if (hipHostMalloc(&h, bytes) != hipSuccess) return nullptr;
  memset(h, 0, bytes);
  p = h;
  return p;
}

struct Blob4K { uint4 v[248]; };   // 3968 bytes: below the 4 KiB kernel-argument limit
__global__ void put_blob_kernel(Blob4K blob, uint4* __restrict__ dst, int n16) {
  if ((int)threadIdx.x < n16) dst[threadIdx.x] = blob.v[threadIdx.x];
}

int upload_table(const void* host, size_t bytes, void* dev, hipStream_t stream) {
  const char* h = (const char*)host;
  char* d = (char*)dev;
  for (size_t off = 0; off < bytes; off += sizeof(Blob4K)) {
    Blob4K blob;
    const size_t n = bytes - off < sizeof(Blob4K) ? bytes - off : sizeof(Blob4K);
    memcpy(&blob, h + off, n);
    hipLaunchKernelGGL(put_blob_kernel, dim3(1), dim3(256), 0, stream, blob, (uint4*)(d + off), (int)((n + 15) / 16));
  }
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

}  // namespace gims

extern "C" int gims_upload_table(const void* host, int64_t bytes, void* dev, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(host && dev && bytes >= 0 && bytes <= (1 << 20) && ((uintptr_t)dev & 15) == 0, "gims_upload_table: bad arguments");
  if (bytes == 0) return GIMS_OK;
  return upload_table(host, (size_t)bytes, dev, (hipStream_t)stream);
}

namespace gims {
// normalize_keypoints (gmatcher.py:26-33) + Conv1d(2->c1) + BN(eval, folded) + ReLU (gmatcher.py:87-97)
template <bool RELU>
__global__ __launch_bounds__(256) void kenc_first_kernel(const float* __restrict__ kpts, const float* __restrict__ norm3,
                                                         const int32_t* __restrict__ seg, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, int c1, float* __restrict__ out,
                                                         int64_t n) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = idx / c1;
  const int ch = (int)(idx - row * c1);
  if (row >= n) return;
  const float* nm = norm3 + 3 * seg[row];
  const float x = (kpts[2 * row] - nm[0]) / nm[2];
  const float y = (kpts[2 * row + 1] - nm[1]) / nm[2];
  float v = fmaf(w1[2 * ch + 1], y, fmaf(w1[2 * ch], x, b1[ch]));
  out[row * c1 + ch] = RELU ? fmaxf(v, 0.f) : v;
}

// LayerNorm of the reference's use_layernorm=True variant (gmatcher.py:74-85) + ReLU: per point (row) over its c channels,
// UNBIASED standard deviation, eps added to the std (not to the variance).  One wave per row, values held in registers
// (c <= 512), two passes (mean, then squared deviations).  Output f32 and/or split-bf16 SPL32 planes for the next GEMM.
__global__ __launch_bounds__(256) void layernorm_act_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int c,
                                                            const float* __restrict__ a2, const float* __restrict__ b2, float eps, int act,
                                                            float* __restrict__ out, int64_t ldo, uint16_t* __restrict__ out_hi,
                                                            uint16_t* __restrict__ out_lo, int64_t ld_split) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  float v[8];
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
    const float d = lane + 64 * k < c ? v[k] - mean : 0.f;
    q += d * d;
  }
  const float sd = sqrtf(wave_sum(q) / (float)(c - 1));
  const float inv = 1.f / (sd + eps);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int ch = lane + 64 * k;
    if (ch < c) {
      float y = a2[ch] * ((v[k] - mean) * inv) + b2[ch];
      if (act == GIMS_ACT_RELU) y = fmaxf(y, 0.f);
      if (out) out[row * ldo + ch] = y;
      if (out_hi) {
        const uint16_t h = f2bf(y);
        out_hi[row * ld_split + spl_col(ch)] = h;
        out_lo[row * ld_split + spl_col(ch)] = f2bf(y - bf2f(h));
      }
    }
  }
}

// out[i,:] = mean over CSR neighbours of h[j,:]   (one wave per node, lanes over channel quads)
__global__ __launch_bounds__(256) void sage_mean_kernel(const float* __restrict__ h, int64_t ldh,
                                                        const int32_t* __restrict__ indptr,
                                                        const int32_t* __restrict__ indices, int n, int c,
                                                        float* __restrict__ out, int64_t ldo,
                                                        uint16_t* __restrict__ out_spl, int64_t ld_spl) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int node = blockIdx.x * 4 + wave;
  if (node >= n) return;
  const int beg = indptr[node], end = indptr[node + 1];
  // The walk was a chain of dependent loads (index, then row, nine times per node: 68 us per launch of 65 536 nodes, the same for 128 and for
  // 256 channels): the indices of up to 64 neighbours are fetched by the lanes at once and handed round by readlane, and four rows are in
  // flight at a time -- added in neighbour order, so the bits are the old ones (round 6).  Control flow is wave-uniform (lanes beyond the
  // channel count read quad 0 and discard): every lane holds its index when another lane asks for it.
  for (int q0 = 0; 4 * q0 < c; q0 += 64) {
    const int q = q0 + lane;
    const bool act = 4 * q < c;
    const int qq = act ? q : 0;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e0 = beg; e0 < end; e0 += 64) {
      const int cnt = end - e0 < 64 ? end - e0 : 64;
      const int mine = lane < cnt ? indices[e0 + lane] : 0;
      int e = 0;
      for (; e + 4 <= cnt; e += 4) {
        const float4 x0 = *(const float4*)(h + (int64_t)__shfl(mine, e, 64) * ldh + 4 * qq);
        const float4 x1 = *(const float4*)(h + (int64_t)__shfl(mine, e + 1, 64) * ldh + 4 * qq);
        const float4 x2 = *(const float4*)(h + (int64_t)__shfl(mine, e + 2, 64) * ldh + 4 * qq);
        const float4 x3 = *(const float4*)(h + (int64_t)__shfl(mine, e + 3, 64) * ldh + 4 * qq);
        s.x += x0.x; s.y += x0.y; s.z += x0.z; s.w += x0.w;
        s.x += x1.x; s.y += x1.y; s.z += x1.z; s.w += x1.w;
        s.x += x2.x; s.y += x2.y; s.z += x2.z; s.w += x2.w;
        s.x += x3.x; s.y += x3.y; s.z += x3.z; s.w += x3.w;
      }
      for (; e < cnt; ++e) {
        const float4 x = *(const float4*)(h + (int64_t)__shfl(mine, e, 64) * ldh + 4 * qq);
        s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
      }
    }
    if (!act) continue;
    // sum / deg, like DGL's mean reducer (a true division, not a multiply by the reciprocal)
    if (end > beg) {
      const float d = (float)(end - beg);
      s.x /= d; s.y /= d; s.z /= d; s.w /= d;
    }
    if (out) *(float4*)(out + (int64_t)node * ldo + 4 * q) = s;
    if (out_spl) {      // SPL32 planes for the split-bf16 GEMM that consumes the mean (a quad never straddles a 32-channel block)
      const uint32_t h01 = pack_bf2(s.x, s.y), h23 = pack_bf2(s.z, s.w);
      const uint32_t l01 = pack_bf2(s.x - __uint_as_float(h01 << 16), s.y - __uint_as_float(h01 & 0xffff0000u));
      const uint32_t l23 = pack_bf2(s.z - __uint_as_float(h23 << 16), s.w - __uint_as_float(h23 & 0xffff0000u));
      uint16_t* o = out_spl + (int64_t)node * ld_spl + spl_col(4 * q);
      *(uint2*)o = make_uint2(h01, h23);
      *(uint2*)(o + 32) = make_uint2(l01, l23);
    }
  }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int64_t lds,
                                                          const int32_t* __restrict__ idx, int n, int c,
                                                          float* __restrict__ dst, int64_t ldd) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= n) return;
  const float* s = src + (int64_t)idx[row] * lds;
  float* d = dst + (int64_t)row * ldd;
  for (int j = lane; j < c; j += 64) d[j] = s[j];
}

// kept-keypoint compaction of a whole batch in one launch (gmatcher.py:244-249): blockIdx.y = image
__global__ __launch_bounds__(256) void pack_graphs_kernel(const gims_pack_image* __restrict__ imgs, int d,
                                                          float* __restrict__ feat, int64_t ldf,
                                                          float* __restrict__ kpts_out, float* __restrict__ score_out,
                                                          int32_t* __restrict__ seg, int32_t* __restrict__ indptr_out,
                                                          int32_t* __restrict__ indices_out, int n_rows_total,
                                                          int n_edges_total) {
  const gims_pack_image im = imgs[blockIdx.y];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // rows: one wave per kept keypoint
  for (int r = blockIdx.x * 4 + wave; r < im.n_kept; r += gridDim.x * 4) {
    const int src = im.kept[r];
    const float4* s = (const float4*)(im.desc + (int64_t)src * im.ldd);
    float4* o = (float4*)(feat + (int64_t)(im.row_off + r) * ldf);
    for (int q = lane; 4 * q < d; q += 64) o[q] = s[q];
    if (lane == 0) {
      kpts_out[2 * (im.row_off + r)] = im.kpts[2 * src];
      kpts_out[2 * (im.row_off + r) + 1] = im.kpts[2 * src + 1];
      if (score_out && im.score) score_out[im.row_off + r] = im.score[src];
      seg[im.row_off + r] = blockIdx.y;
      indptr_out[im.row_off + r] = im.indptr[r] + im.edge_off;
    }
  }
  // edges
  for (int e = blockIdx.x * 256 + threadIdx.x; e < im.n_edges; e += gridDim.x * 256)
    indices_out[im.edge_off + e] = im.indices[e] + im.row_off;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) indptr_out[n_rows_total] = n_edges_total;
}

// channel-major (D,N) descriptors of a whole batch -> one point-major [rows][D] buffer (64x64 LDS tile transpose),
// plus the keypoint / score rows; blockIdx.z = image
__global__ __launch_bounds__(256) void ingest_kernel(const gims_ingest_image* __restrict__ imgs, int d,
                                                     float* __restrict__ desc_out, int64_t ldo,
                                                     float* __restrict__ kpts_out, float* __restrict__ score_out) {
  __shared__ float tile[64][65];
  const gims_ingest_image im = imgs[blockIdx.z];
  const int n0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
  if (n0 >= im.n) return;
  const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = 4 * i + r4;                       // channel within the tile
    const int n = n0 + c;
    tile[r][c] = (d0 + r < d && n < im.n) ? im.desc[(int64_t)(d0 + r) * im.ldd + n] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = 4 * i + r4;                       // keypoint within the tile
    const int n = n0 + r;
    if (n < im.n && d0 + c < d) desc_out[(int64_t)(im.row_off + n) * ldo + d0 + c] = tile[c][r];
  }
  if (blockIdx.y == 0 && threadIdx.x < 64) {
    const int n = n0 + threadIdx.x;
    if (n < im.n) {
      kpts_out[2 * (im.row_off + n)] = im.kpts[2 * n];
      kpts_out[2 * (im.row_off + n) + 1] = im.kpts[2 * n + 1];
      if (score_out) score_out[im.row_off + n] = im.score ? im.score[n] : 0.f;
    }
  }
}

}  // namespace gims

extern "C" int gims_ingest_images(const gims_ingest_image* dev_images, int32_t n_images, int32_t max_n, int32_t d,
                                  float* desc_out, int64_t ldo, float* kpts_out, float* score_out, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(dev_images && n_images > 0 && max_n > 0 && d > 0 && desc_out && kpts_out, "gims_ingest_images: bad arguments");
  hipLaunchKernelGGL(ingest_kernel, dim3(cdiv(max_n, 64), cdiv(d, 64), n_images), dim3(256), 0, (hipStream_t)stream,
                     dev_images, d, desc_out, ldo, kpts_out, score_out);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_pack_graphs(const gims_pack_image* dev_images, int32_t n_images, int32_t max_kept, int32_t max_edges,
                                int32_t d, float* feat, int64_t ldf, float* kpts_out, float* score_out, int32_t* seg,
                                int32_t* indptr_out, int32_t* indices_out, int32_t n_rows_total, int32_t n_edges_total,
                                void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(dev_images && n_images > 0 && feat && kpts_out && seg && indptr_out && indices_out, "gims_pack_graphs: null pointer");
  GIMS_CHECK_ARG((d % 4) == 0 && (ldf % 4) == 0, "gims_pack_graphs: d / ldf must be multiples of 4");
  int gx = cdiv(max_kept, 4);
  const int ge = cdiv(max_edges, 256);
  gx = gx > ge ? gx : ge;
  if (gx < 1) gx = 1;
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(pack_graphs_kernel, dim3(gx, n_images), dim3(256), 0, (hipStream_t)stream, dev_images, d, feat, ldf,
                     kpts_out, score_out, seg, indptr_out, indices_out, n_rows_total, n_edges_total);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_run_ops(const gims_op* ops, int32_t n_ops, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(ops && n_ops >= 0, "gims_run_ops: bad arguments");
  for (int i = 0; i < n_ops; ++i) {
    int rc;
    if (ops[i].kind == GIMS_OP_LINEAR) {
      rc = gims_linear(&ops[i].u.lin, stream);
    } else if (ops[i].kind == GIMS_OP_ATTENTION) {
      rc = gims_attention_ex(&ops[i].u.att, stream);
    } else if (ops[i].kind == GIMS_OP_AUX) {
      const gims_aux_args& x = ops[i].u.aux;
      if (x.fn == GIMS_AUX_SPLIT_SPL32)
        rc = gims_split_spl32((const float*)x.p[0], x.i[0], (uint16_t*)x.p[1], x.i[1], x.i[2], (int32_t)x.i[3], stream);
      else if (x.fn == GIMS_AUX_SAGE_MEAN_SPLIT)
        rc = gims_sage_mean_split((const float*)x.p[0], x.i[0], (const int32_t*)x.p[1], (const int32_t*)x.p[2], (int32_t)x.i[1], (int32_t)x.i[2],
                                  (uint16_t*)x.p[3], x.i[3], stream);
      else if (x.fn == GIMS_AUX_KENC_FIRST)
        rc = gims_kenc_first((const float*)x.p[0], (const float*)x.p[1], (const int32_t*)x.p[2], (const float*)x.p[3], (const float*)x.p[4],
                             (int32_t)x.i[0], (float*)x.p[5], x.i[1], stream);
      else
        GIMS_CHECK_ARG(false, "gims_run_ops: op %d: unknown auxiliary function %d", i, x.fn);
    } else {
      GIMS_CHECK_ARG(false, "gims_run_ops: op %d has unknown kind %d", i, ops[i].kind);
    }
    if (rc != GIMS_OK) return rc;
  }
  return GIMS_OK;
}

// The same replay with a HIP event recorded on `stream` before the first op and after every op (events[0 .. n_ops]): the
// per-kernel durations of the production path, measured on the stream the kernels are launched on.
extern "C" int gims_run_ops_timed(const gims_op* ops, int32_t n_ops, void* stream, void* const* events) {
  using namespace gims;
  GIMS_CHECK_ARG(ops && n_ops >= 0 && events, "gims_run_ops_timed: bad arguments");
  GIMS_HIP(hipEventRecord((hipEvent_t)events[0], (hipStream_t)stream));
  for (int i = 0; i < n_ops; ++i) {
    const int rc = gims_run_ops(ops + i, 1, stream);
    if (rc != GIMS_OK) return rc;
    GIMS_HIP(hipEventRecord((hipEvent_t)events[i + 1], (hipStream_t)stream));
  }
  return GIMS_OK;
}

extern "C" int gims_events_create(int32_t n, void** events_out) {
  using namespace gims;
  GIMS_CHECK_ARG(n > 0 && events_out, "gims_events_create: bad arguments");
  for (int i = 0; i < n; ++i) {
    hipEvent_t e = nullptr;
    const hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) {
      for (int j = 0; j < i; ++j) (void)hipEventDestroy((hipEvent_t)events_out[j]);
      GIMS_HIP(rc);
    }
    events_out[i] = (void*)e;
  }
  return GIMS_OK;
}

extern "C" int gims_events_record(void* event, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(event, "gims_events_record: null event");
  GIMS_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return GIMS_OK;
}

// ms_out[i] = time between events[i] and events[i+1], i < n - 1; every event must have completed (synchronise first).
extern "C" int gims_events_elapsed(void* const* events, int32_t n, float* ms_out) {
  using namespace gims;
  GIMS_CHECK_ARG(events && n > 1 && ms_out, "gims_events_elapsed: bad arguments");
  for (int i = 0; i + 1 < n; ++i) GIMS_HIP(hipEventElapsedTime(&ms_out[i], (hipEvent_t)events[i], (hipEvent_t)events[i + 1]));
  return GIMS_OK;
}

extern "C" int gims_events_destroy(void* const* events, int32_t n) {
  using namespace gims;
  GIMS_CHECK_ARG(events && n >= 0, "gims_events_destroy: bad arguments");
  for (int i = 0; i < n; ++i)
    if (events[i]) GIMS_HIP(hipEventDestroy((hipEvent_t)events[i]));
  return GIMS_OK;
}

extern "C" int gims_ops_graph_create(const gims_op* ops, int32_t n_ops, void* stream, void** graph_exec_out) {
  using namespace gims;
  GIMS_CHECK_ARG(ops && n_ops > 0 && graph_exec_out && stream, "gims_ops_graph_create: bad arguments (a non-default stream is required)");
  hipStream_t s = (hipStream_t)stream;
  GIMS_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  const int rc = gims_run_ops(ops, n_ops, stream);
  hipGraph_t graph = nullptr;
  const hipError_t e = hipStreamEndCapture(s, &graph);
  if (rc != GIMS_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  GIMS_HIP(e);
  hipGraphExec_t exec = nullptr;
  const hipError_t e2 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  GIMS_HIP(e2);
  *graph_exec_out = (void*)exec;
  return GIMS_OK;
}

extern "C" int gims_ops_graph_launch(void* graph_exec, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(graph_exec, "gims_ops_graph_launch: null graph");
  GIMS_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return GIMS_OK;
}

extern "C" int gims_ops_graph_destroy(void* graph_exec) {
  using namespace gims;
  if (graph_exec) GIMS_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return GIMS_OK;
}

extern "C" int gims_abi_version(void) { return GIMS_ABI_VERSION; }
extern "C" const char* gims_last_error(void) { return gims::g_err; }
extern "C" int gims_stream_sync(void* stream) {
  using namespace gims;
  GIMS_HIP(hipStreamSynchronize((hipStream_t)stream));
  return GIMS_OK;
}

extern "C" int gims_kenc_first(const float* kpts, const float* norm3, const int32_t* seg_of_row, const float* w1,
                               const float* b1, int32_t c1, float* out, int64_t n, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(kpts && norm3 && seg_of_row && w1 && b1 && out && c1 > 0 && n >= 0, "gims_kenc_first: bad arguments");
  if (n == 0) return GIMS_OK;
  hipLaunchKernelGGL(kenc_first_kernel<true>, dim3(cdiv(n * c1, 256)), dim3(256), 0, (hipStream_t)stream, kpts, norm3,
                     seg_of_row, w1, b1, c1, out, n);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_kenc_first_linear(const float* kpts, const float* norm3, const int32_t* seg_of_row, const float* w1,
                                      const float* b1, int32_t c1, float* out, int64_t n, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(kpts && norm3 && seg_of_row && w1 && b1 && out && c1 > 0 && n >= 0, "gims_kenc_first_linear: bad arguments");
  if (n == 0) return GIMS_OK;
  hipLaunchKernelGGL(kenc_first_kernel<false>, dim3(cdiv(n * c1, 256)), dim3(256), 0, (hipStream_t)stream, kpts, norm3,
                     seg_of_row, w1, b1, c1, out, n);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_layernorm_act(const float* x, int64_t ldx, int64_t rows, int32_t c, const float* a2, const float* b2, float eps,
                                  int32_t act, float* out, int64_t ldo, uint16_t* out_hi, uint16_t* out_lo, int64_t ld_split,
                                  void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(x && a2 && b2 && (out || out_hi) && rows >= 0 && c >= 2 && c <= 512, "gims_layernorm_act: bad arguments (2 <= c <= 512)");
  GIMS_CHECK_ARG((out_hi == nullptr) == (out_lo == nullptr), "gims_layernorm_act: out_hi and out_lo come together");
  if (rows == 0) return GIMS_OK;
  hipLaunchKernelGGL(layernorm_act_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, c, a2, b2, eps,
                     act, out, ldo, out_hi, out_lo, ld_split);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_sage_mean(const float* h, int64_t ldh, const int32_t* indptr, const int32_t* indices, int32_t n,
                              int32_t c, float* out, int64_t ldo, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(h && indptr && indices && out && n >= 0, "gims_sage_mean: bad arguments");
  GIMS_CHECK_ARG((c % 4) == 0 && (ldh % 4) == 0 && (ldo % 4) == 0, "gims_sage_mean: c / ld must be multiples of 4");
  if (n == 0) return GIMS_OK;
  hipLaunchKernelGGL(sage_mean_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, h, ldh, indptr, indices, n, c,
                     out, ldo, (uint16_t*)nullptr, (int64_t)0);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_sage_mean_split(const float* h, int64_t ldh, const int32_t* indptr, const int32_t* indices, int32_t n,
                                    int32_t c, uint16_t* out_spl, int64_t ld_spl, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(h && indptr && indices && out_spl && n >= 0, "gims_sage_mean_split: bad arguments");
  GIMS_CHECK_ARG((c % 4) == 0 && (ldh % 4) == 0 && (ld_spl % 4) == 0 && ld_spl >= 2 * (int64_t)((c + 31) / 32 * 32) && ((uintptr_t)out_spl & 7) == 0,
                 "gims_sage_mean_split: c / ldh multiples of 4, SPL32 output pitch >= 2 * c rounded up to 32 channels");
  if (n == 0) return GIMS_OK;
  hipLaunchKernelGGL(sage_mean_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, h, ldh, indptr, indices, n, c,
                     (float*)nullptr, (int64_t)0, out_spl, ld_spl);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

// per-pair match statistics: one workgroup per pair, fixed summation order (deterministic)
__global__ __launch_bounds__(256) void pair_stats_kernel(const int64_t* __restrict__ matches0, const float* __restrict__ scores0,
                                                         const int32_t* __restrict__ tab, float* __restrict__ out) {
  const int p = blockIdx.x, t = threadIdx.x;
  const int n0 = tab[4 * p + 1], off = tab[4 * p + 3];
  __shared__ float ssum[256];
  __shared__ int scnt[256];
  float s = 0.f;
  int c = 0;
  for (int i = t; i < n0; i += 256)
    if (matches0[off + i] >= 0) { s += scores0[off + i]; ++c; }
  ssum[t] = s;
  scnt[t] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) { ssum[t] += ssum[t + o]; scnt[t] += scnt[t + o]; }
    __syncthreads();
  }
  if (t == 0) {
    float* r = out + 5 * p;
    r[0] = (float)tab[4 * p]; r[1] = (float)n0; r[2] = (float)tab[4 * p + 2];
    r[3] = (float)scnt[0]; r[4] = ssum[0] / fmaxf((float)scnt[0], 1.f);
  }
}

extern "C" int gims_pair_stats(const int64_t* matches0, const float* scores0, const int32_t* table, int32_t n_pairs, float* out,
                               void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(matches0 && scores0 && table && out && n_pairs >= 0, "gims_pair_stats: bad arguments");
  if (n_pairs == 0) return GIMS_OK;
  hipLaunchKernelGGL(pair_stats_kernel, dim3(n_pairs), dim3(256), 0, (hipStream_t)stream, matches0, scores0, table, out);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

extern "C" int gims_gather_rows(const float* src, int64_t lds, const int32_t* idx, int32_t n, int32_t c, float* dst,
                                int64_t ldd, void* stream) {
  using namespace gims;
  GIMS_CHECK_ARG(src && idx && dst && n >= 0 && c > 0, "gims_gather_rows: bad arguments");
  if (n == 0) return GIMS_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, src, lds, idx, n, c, dst, ldd);
  GIMS_LAUNCH_CHECK();
  return GIMS_OK;
}

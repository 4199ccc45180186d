/*
 * gims_hip.h -- C ABI of libgims_hip.so: the MI355X (gfx950) kernels behind the GIMS matcher hot path.
 *
 * The reference (songxf1024/GIMS) has no FFI boundary of its own for this path: the boundary is the
 * Python nn.Module API  GMatcher(config).forward(data)  (models/gmatcher.py:177,219) and
 * Matching(config).forward(data)  (models/matching.py:10,15).  gims_amd/gmatcher.py keeps that API and
 * calls the entry points below through ctypes; each entry point names the reference lines it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless its name starts with h_ (host);
 *   - every call enqueues work on `stream` (a hipStream_t passed as void*) and returns immediately,
 *     except the two calls documented as synchronising;
 *   - return value: 0 on success, a negative GIMS_E* code on failure (never throws across the ABI);
 *     gims_last_error() returns a thread-local message for the last failure;
 *   - activations are POINT-MAJOR: row = keypoint, column = channel (the reference is channel-major
 *     (B,C,N); the Python shell transposes at the boundary);
 *   - no hidden global state; safe to call from several threads on different streams.
 */
#ifndef GIMS_HIP_H
#define GIMS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GIMS_ABI_VERSION 2   /* 2 (round 6): gims_attn_guard grew `max_thr` (so did gims_linear_args / gims_attn_args, which embed it); + gims_attention_launch_counts, gims_agc_workspace_bytes_ex */

#define GIMS_OK 0
#define GIMS_EINVAL (-1)   /* bad argument (shape / alignment / null pointer) */
#define GIMS_EHIP (-2)     /* a HIP runtime call failed */
#define GIMS_ENUMERIC (-3) /* numeric guard tripped (non-finite Sinkhorn marginal) */

int gims_abi_version(void);
const char* gims_last_error(void);
/* Blocks until `stream` is idle (hipStreamSynchronize). */
int gims_stream_sync(void* stream);
/* Copies a small host table (descriptor arrays, offsets, <= 1 MiB) to device memory IN STREAM ORDER without touching the
 * copy engines: the bytes travel as kernel arguments (chunks of 3968 B).  Unlike a pageable hipMemcpy this never blocks
 * the submitting thread on the stream and never pins pages, so the host can keep running ahead of the GPU.  `dev` must be
 * 16-byte aligned and padded to a multiple of 16 bytes.  No reference counterpart (host-side plumbing of this build). */
int gims_upload_table(const void* host, int64_t bytes, void* dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Linear layers (1x1 Conv1d / nn.Linear):   C[m, n] = act( sum_k A[m,k] * W[n,k] + bias[n] ) (+ R[m,n])
 * replaces: nn.Conv1d(k=1) stacks of MLP / MultiHeadedAttention / final_proj (gmatcher.py:11-24,
 * 106-125, 202-205, 273), SAGEConv's fc_self / fc_neigh (gmatcher.py:149-151) and the two dense
 * contractions of the path: cosine similarity D D^T (agc.py:390) and the score matrix
 * einsum('bdn,bdm->bnm') (gmatcher.py:274).
 * A is given as up to two K-segments (concat-free torch.cat([x, message]), gmatcher.py:125):
 *   k <  k0 : a0[m*lda0 + k]          k >= k0 : a1[m*lda1 + (k-k0)]
 * W is row-major [n][K] (PyTorch weight layout).  K and k0 must be multiples of 32.
 * precision: GIMS_PREC_F32   -> exact-f32 MFMA (v_mfma_f32_32x32x2_f32), f32 operands in a0/a1/w;
 *            GIMS_PREC_BF16X3-> split-bf16 MFMA (hi*hi + hi*lo + lo*hi); W comes pre-split in
 *                               w (hi plane) / w_lo (lo plane), A is split on the fly.
 * out_f32 / out_bf16 / (out_hi,out_lo) may each be NULL; `residual` (f32, same ld as out_f32) may alias out_f32.
 */
#define GIMS_PREC_F32 0
#define GIMS_PREC_BF16X3 1
#define GIMS_PREC_BF16X6 2   /* three-way split (a1+a2+a3, exact), six bf16 MFMAs per product: f32-GEMM accuracy at 6/16 of
                              the exact-f32 MFMA cost.  Operands in the SPL3 layout (gims_split_spl3); batched launches only
                              (gims_linear_put_many + gims_linear_batch), C = scale * A W^T into out_f32, GIMS_LINEAR_UPPER ok */
#define GIMS_ACT_NONE 0
#define GIMS_ACT_RELU 1

/* Device-side guard of a launch (attention_precision='auto' of the host shell: a layer that ran on cheap operands is REDONE at f32-class accuracy
 * inside the same batch when the statistic of the cheap launch says the operands did not suffice -- no host round trip, so the results of a
 * batch never leave with the cheap tier's error).  A launch whose args carry a guard with stat != NULL is a no-op unless the guard FIRES; every
 * workgroup evaluates it at entry from `stat`, the accumulator a preceding gims_attention_stat launch of the same stream filled:
 *   GIMS_GUARD_PEAKED: some head's mean row maximum (stat[h][0] / stat[h][1] / 2^24) exceeds mean_thr, or its share of rows with a maximum above
 *                      1/2 (stat[h][3] / stat[h][1]) exceeds tail_thr, or -- max_thr > 0 -- its LARGEST row maximum (stat[h][2] / 2^24) reaches
 *                      max_thr: a single sharply peaked row inside a diffuse layer (round 6)        (guards a plain-bf16 attention layer);
 *   GIMS_GUARD_RANGE : max |Q|, |K| or |V| as stored (stat[n_heads][0..2]) exceeds range_limit, or is not finite   (guards an IEEE-half layer).
 * A guarded gims_attention launch that fires stores 1 into stat[n_heads][3] (the host's record that the layer was redone).  The comparisons
 * are done in float64 exactly as written here, so a host that reads `stat` back reaches the same verdict.  stat == NULL: unconditional. */
#define GIMS_GUARD_PEAKED 1
#define GIMS_GUARD_RANGE 2
typedef struct gims_attn_guard {
  uint64_t* stat;                        /* [n_heads + 1][4], see gims_attention_stat; NULL = no guard */
  double mean_thr, tail_thr, range_limit;
  int32_t n_heads, kind;                 /* GIMS_GUARD_* */
  double max_thr;                        /* GIMS_GUARD_PEAKED: 0 = the largest row maximum is not looked at */
} gims_attn_guard;

typedef struct gims_linear_args {
  const float* a0; int64_t lda0;
  const float* a1; int64_t lda1;       /* may be NULL when k0 == K */
  const void* w; const void* w_lo; int64_t ldw;  /* f32 (F32) or bf16 planes (BF16X3) */
  const float* bias;                   /* [n] or NULL */
  const float* residual;               /* [m][ldc] or NULL */
  float* out_f32; int64_t ldc;         /* may be NULL */
  uint16_t* out_bf16; int64_t ldc_bf16;/* may be NULL */
  int32_t m, n, k, k0;
  int32_t act;                         /* GIMS_ACT_* */
  int32_t precision;                   /* GIMS_PREC_* */
  float scale;                         /* applied to the accumulator before bias (1.0 = none) */
  /* pre-split operands (hot path of the attentional GNN): when a0_lo != NULL, a0 (and a1) and w are bf16 buffers in
   * the SPL32 layout written by the producing kernel / gims_split_spl32: logical [rows][K] stored as [rows][pitch >= 2K],
   * per row and 32-channel block 32 hi values then 32 lo values (hi = bf16(x), lo = bf16(x - hi)); channel k sits at
   * (k/32)*64 + k%32 (hi) and +32 (lo).  a0_lo = a0 + 32, w_lo = w + 32 (the kernel only uses them as flags), lda/ldw
   * are the pitches, 128-byte aligned.  precision must be GIMS_PREC_BF16X3.  Not available through gims_linear_batch. */
  const uint16_t* a0_lo; const uint16_t* a1_lo;
  /* optional split output in the SPL32 layout: out_hi = base, out_lo = base + 32, ld_split = pitch (>= 2n) */
  uint16_t* out_hi; uint16_t* out_lo; int64_t ld_split;
  /* GIMS_LINEAR_UPPER: symmetric product (A == W): skip output tiles that lie entirely below the diagonal */
  int32_t flags;
  /* GIMS_LINEAR_CONV3 (pre-split operands only): the GEMM is a 3x3 convolution (pad 1) over an NHWC activation held in
   * the SPL32 layout -- a0 = [patches * conv_h * conv_w][2C] with pitch lda0, C % 32 == 0, k = 9 C, W columns ordered
   * (ky*3+kx)*C + c.  Output row r is output pixel (patch, yo, xo) of the [(conv_h-1)/stride+1] x [(conv_w-1)/stride+1]
   * grid, and K block (tap, c0..c0+31) of its operand row is read straight from input pixel
   * (yo*stride + ky - 1, xo*stride + kx - 1), or from the 128 zero bytes at a1 when that lies outside: no im2col buffer. */
  int32_t conv_h, conv_w, conv_stride, conv_reserved;
  gims_attn_guard guard;               /* pre-split operands (a0_lo != NULL) only: the launch is a no-op unless the guard fires; zero = always run */
  /* pre-split operands with GIMS_LINEAR_OUT_F16 only: the launch also reports max |value| of what it stores into out_bf16, per block of
   * 256 output columns (Q | K | V of the projection in front of a GIMS_ATTN_F16 launch): range_stat[b] = max(range_stat[b], float bits of
   * the maximum) for column block b = col / 256 < 3, by integer atomics on the f32 bit patterns -- the range row of gims_attention_stat's
   * accumulator (pass stat + 4 * n_heads), measured where the values are produced instead of by a scan of the buffer.  NULL: not measured. */
  uint64_t* range_stat;
} gims_linear_args;
#define GIMS_LINEAR_UPPER 1
  /* GIMS_LINEAR_HI_ONLY (pre-split operands): multiply the hi planes only -- a plain bf16 product (2^-9 relative per
   * operand) at a third of the matrix work, for results that are rounded to bf16 anyway (the Q/K/V projection) */
#define GIMS_LINEAR_HI_ONLY 2
  /* (flag value 4 was GIMS_LINEAR_A1_HI_ONLY until round 5: the hi-planes-only product for the second A segment -- the attention message in MLP0 --
   * measured +1.5 % pairs/s at 6.4e-5 of the 1e-4 score bar and was removed) */
#define GIMS_LINEAR_CONV3 8
  /* GIMS_LINEAR_OUT_F16: out_bf16 receives IEEE half instead of bf16 (round to nearest even, saturated to +-65504 so that no
   * infinity is ever stored) -- the Q/K/V projection in front of the GIMS_ATTN_F16 attention kernels */
#define GIMS_LINEAR_OUT_F16 16

int gims_linear(const gims_linear_args* args, void* stream);
/* Many independent problems in ONE launch (ragged batch: per-pair score matrices, per-image similarity
 * matrices).  gims_linear_put validates one descriptor and stores it into device memory (by-value kernel
 * argument: no host staging copy, no synchronisation); gims_linear_batch launches all `count` problems,
 * grid sized for the largest (max_m x max_n).  All problems of a batch share `precision`. */
int gims_linear_put(const gims_linear_args* args, gims_linear_args* dev_dst, void* stream);
int gims_linear_put_many(const gims_linear_args* h_args /* HOST array */, int32_t count, gims_linear_args* dev_dst, void* stream);
int gims_linear_batch(const gims_linear_args* dev_args, int32_t count, int32_t max_m, int32_t max_n,
                      int32_t precision, void* stream);

/* Split an f32 array into bf16 hi/lo planes (hi = bf16_rne(x), lo = bf16_rne(x - hi)). */
int gims_split_bf16(const float* src, uint16_t* hi, uint16_t* lo, int64_t n, void* stream);
/* f32 [rows][k] (pitch lds) -> SPL32 bf16 [rows][2k] (pitch ldd); k % 32 == 0. */
/* f32 [rows][k] (row pitch lds) -> SPL3 bf16 [rows][3k] (row pitch ldd elements): per 32-channel block 32 x a1, 32 x a2,
 * 32 x a3 with a = a1 + a2 + a3 exactly.  Operand format of GIMS_PREC_BF16X6. */
int gims_split_spl3(const float* src, int64_t lds, uint16_t* dst, int64_t ldd, int64_t rows, int32_t k, void* stream);
int gims_split_spl32(const float* src, int64_t lds, uint16_t* dst, int64_t ldd, int64_t rows, int32_t k, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-head attention message   O = softmax(Q K^T / sqrt(dh)) V   per head, flash-style.
 * replaces: attention() + the head view of MultiHeadedAttention.forward (gmatcher.py:35-39,109-113).
 * qkv: bf16 [rows][ld] with, per row, Q at column q_col + h*64 + d, K at k_col + ..., V at v_col + ...
 *      (head-BLOCKED channels; the Python shell permutes the reference's interleaved heads
 *      (view(B, dh, H, N), gmatcher.py:111) into the projection weights).
 * Each problem p attends queries [q_off, q_off+n_q) to keys/values [kv_off, kv_off+n_kv).
 * out: f32 [rows][ld_out], head-blocked columns h*64 + d.       dh = 64, heads = n_heads.
 * out_hi/out_lo: the same result in the SPL32 split-bf16 layout (out_lo = out_hi + 32, ld_split = pitch >= 512) for
 * the pre-split linear that consumes the message.
 * flags: GIMS_ATTN_Q_PRESCALED = the caller already multiplied Q by log2(e)/sqrt(dh) (e.g. folded into the rows of the
 * query projection): the kernel then computes softmax as exp2(Q K^T) / sum, which saves it one multiply-add per score.
 */
#define GIMS_ATTN_Q_PRESCALED 1
/* GIMS_ATTN_X3: f32-class accuracy for sharply peaked softmaxes.  qkv is then the SPL32 split-bf16 buffer the 3-pass Q/K/V
 * projection writes (gims_linear out_hi/out_lo; pitch ld >= 2 * 768, see gims_linear_args), q_col/k_col/v_col stay LOGICAL
 * channel offsets (multiples of 32), and every product of the kernel (Q K^T and P V) is three bf16 MFMAs on hi/lo pairs
 * (hi*hi + hi*lo + lo*hi); P is split in registers.  About three times the matrix work of the plain bf16 kernel. */
#define GIMS_ATTN_X3 2
/* GIMS_ATTN_F16: qkv holds IEEE half instead of bf16 (same layout; written by gims_linear with GIMS_LINEAR_OUT_F16 from the 3-pass
 * projection) and the kernels multiply on v_mfma_f32_32x32x16_f16 -- the bf16 kernels' structure and rate with three more mantissa
 * bits (2^-12 instead of 2^-9 relative per operand), P rounded to half with a row reference that keeps it below 2^15.  For
 * peaked softmaxes, where bf16 operands miss the reference's 1e-4 score bar; |Q|, |K|, |V| must stay below 65504 (the statistic of
 * gims_attention_stat reports them).  Not together with GIMS_ATTN_X3. */
#define GIMS_ATTN_F16 4
/* GIMS_ATTN_NO_RANGE: a measured launch (stat != NULL) leaves the range row of the accumulator alone -- the projection in front of it reported it
 * (gims_linear_args.range_stat), or the caller has no use for it */
#define GIMS_ATTN_NO_RANGE 8
typedef struct gims_attn_problem { int32_t q_off, n_q, kv_off, n_kv; } gims_attn_problem;

int gims_attention(const uint16_t* qkv, int64_t ld, int32_t q_col, int32_t k_col, int32_t v_col,
                   const gims_attn_problem* problems /* device */, int32_t n_problems, int32_t max_n_q,
                   int32_t n_heads, float* out /* may be NULL */, int64_t ld_out,
                   uint16_t* out_hi /* may be NULL */, uint16_t* out_lo, int64_t ld_split, int32_t flags, void* stream);
/* The same launch, additionally reporting how PEAKED the softmax rows were -- the quantity that decides whether plain bf16
 * operands keep the reference's 1e-4 score bar (models/gmatcher.py:35-39 computes the softmax in f32): stat is a device array
 * [n_heads + 1][4] of uint64.  Rows 0 .. n_heads-1: {sum over the reported queries of max_k P[q,k] in 2^-24 fixed point, number of
 * reported queries, largest row maximum (same fixed point), number of reported queries with a row maximum above 1/2}; row n_heads:
 * {bit patterns of max|Q|, max|K|, max|V| as stored (f32), unused} over the rows the launch touches -- the range guard of
 * GIMS_ATTN_F16, filled by GIMS_ATTN_F16 and GIMS_ATTN_X3 launches only (bf16 operands have f32's range).  ACCUMULATED with integer atomics (zero it first; order-independent).
 * The "largest row maximum" of a head is COMPLETE for every kernel (round 6): the 8-wave bf16 kernel, whose other figures come from the sample,
 * adds for EVERY query an upper bound of its row maximum whenever that bound reaches 1/2 -- the largest share of the row's mass that fell into
 * one 32-key half tile (problems of at least 512 keys; a workgroup that had to repeat its tiles in the exact pass reports 1) -- so a single
 * sharply peaked row cannot hide behind the sample.
 * The running-maximum kernels (GIMS_ATTN_X3, the 4-wave and split-key bf16 kernels) report every query as a by-product; the
 * 8-wave bf16 kernel tracks no maximum, so for launches it serves a second, small kernel measures 32 evenly spaced queries of
 * every (problem, head) against all keys (a few microseconds).  stat == NULL: exactly gims_attention. */
int gims_attention_stat(const uint16_t* qkv, int64_t ld, int32_t q_col, int32_t k_col, int32_t v_col,
                        const gims_attn_problem* problems /* device */, int32_t n_problems, int32_t max_n_q,
                        int32_t n_heads, float* out /* may be NULL */, int64_t ld_out,
                        uint16_t* out_hi /* may be NULL */, uint16_t* out_lo, int64_t ld_split, int32_t flags,
                        uint64_t* stat /* device, may be NULL */, void* stream);

/* Which kernel served the attention launches of this process so far (host-side counters, one per kernel family; thread-safe):
 * counts[k] for k < min(n, GIMS_ATTN_KERNEL_KINDS), the rest zero; reset != 0 clears them afterwards.  The launcher picks a kernel
 * from the launch shape (a batch that fills the chip takes the 8-wave kernel, one pair alone the split-key / 4-wave kernels):
 * parity tests use this to assert that a fixture went through the kernel it is meant to pin.  Test / diagnostic hook. */
#define GIMS_ATTN_KERNEL_WAVE4 0       /* attention_bf16_kernel<QP, F16>: 4 waves, running maximum */
#define GIMS_ATTN_KERNEL_SPLIT 1       /* attention_split_kernel<NS, F16>: key range split over wave groups */
#define GIMS_ATTN_KERNEL_WAVE8 2       /* attention8_bf16_kernel, bf16 operands (the kernel of the timed batches) */
#define GIMS_ATTN_KERNEL_WAVE8_F16 3   /* attention8_bf16_kernel, IEEE-half operands */
#define GIMS_ATTN_KERNEL_X3 4          /* attention_x3_kernel / attention_x3w_kernel, unguarded */
#define GIMS_ATTN_KERNEL_X3_GUARDED 5  /* the same behind a gims_attn_guard (the device-side redo of 'auto') */
#define GIMS_ATTN_KERNEL_KINDS 6
int gims_attention_launch_counts(uint64_t* counts /* host */, int32_t n, int32_t reset);

/* gims_attention_stat with its arguments in a struct, plus the guard (see gims_attn_guard): what an op of gims_run_ops executes. */
struct gims_attn_args;
int gims_attention_ex(const struct gims_attn_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * A recorded sequence of launches replayed by ONE call: the 18 layers of AttentionalGNN.forward (gmatcher.py:127-143) are
 * 72 launches whose arguments only change when the batch geometry does, and a caller in an interpreted language pays for
 * every crossing of the ABI.  ops: HOST array; each op is exactly one gims_linear, gims_attention or (GIMS_OP_AUX) small-kernel call, in order, on
 * `stream`.  Stops at (and returns) the first error.
 */
typedef struct gims_attn_args {
  const uint16_t* qkv; int64_t ld; int32_t q_col, k_col, v_col;
  const gims_attn_problem* problems; int32_t n_problems, max_n_q, n_heads;
  float* out; int64_t ld_out; uint16_t* out_hi; uint16_t* out_lo; int64_t ld_split; int32_t flags;
  uint64_t* stat;                      /* gims_attention_stat's peakedness accumulator, or NULL */
  gims_attn_guard guard;               /* GIMS_ATTN_X3 launches only: no-op unless the guard fires; zero = always run */
} gims_attn_args;
#define GIMS_OP_LINEAR 0
#define GIMS_OP_ATTENTION 1
/* GIMS_OP_AUX (round 6): the small kernels of the encoder stage in front of the layers (GraphSAGE, gmatcher.py:145-162, 268-269; keypoint encoder,
 * gmatcher.py:87-97, 270-271), so that that stage replays from a table like the layers do -- between the one host synchronisation of a batch and
 * the layers the device waits for the host, and a dozen calls across the ABI were most of that wait.  fn selects the entry point, p / i are its
 * pointer and integer arguments in declaration order:
 *   GIMS_AUX_SPLIT_SPL32      gims_split_spl32     p = {src, dst}                              i = {lds, ldd, rows, k}
 *   GIMS_AUX_SAGE_MEAN_SPLIT  gims_sage_mean_split p = {h, indptr, indices, out_spl}           i = {ldh, n, c, ld_spl}
 *   GIMS_AUX_KENC_FIRST       gims_kenc_first      p = {kpts, norm3, seg_of_row, w1, b1, out}  i = {c1, n} */
#define GIMS_OP_AUX 2
#define GIMS_AUX_SPLIT_SPL32 0
#define GIMS_AUX_SAGE_MEAN_SPLIT 1
#define GIMS_AUX_KENC_FIRST 2
typedef struct gims_aux_args { int32_t fn, reserved; const void* p[6]; int64_t i[4]; } gims_aux_args;
typedef struct gims_op { int32_t kind; int32_t reserved; union { gims_linear_args lin; gims_attn_args att; gims_aux_args aux; } u; } gims_op;
int gims_run_ops(const gims_op* ops /* HOST */, int32_t n_ops, void* stream);
/* The same replay with a HIP event recorded on `stream` before the first op and after every op: events is a HOST array of
 * n_ops + 1 event handles from gims_events_create.  gims_events_elapsed (after the stream has been synchronised) returns the
 * n - 1 intervals between n consecutive events in milliseconds: per-kernel durations of the production path, taken on the
 * stream the kernels run on.  (Instrumentation of this build; the reference prints wall-clock stage times, gmatcher.py:226-243.) */
int gims_run_ops_timed(const gims_op* ops /* HOST */, int32_t n_ops, void* stream, void* const* events /* HOST, n_ops + 1 */);
int gims_events_create(int32_t n, void** events_out /* HOST array of n handles */);
int gims_events_record(void* event, void* stream);
int gims_events_elapsed(void* const* events /* HOST */, int32_t n, float* h_ms_out /* HOST, n - 1 */);
int gims_events_destroy(void* const* events /* HOST */, int32_t n);
/* The same sequence as a HIP graph: gims_ops_graph_create captures the launches of `ops` on `stream` (nothing executes),
 * gims_ops_graph_launch replays them (one graph launch: no per-kernel launch gaps), gims_ops_graph_destroy frees the graph.
 * Every op must have run once through gims_run_ops before (first-use initialisation cannot happen inside a capture), and
 * every pointer in `ops` must stay valid for as long as the graph is launched. */
int gims_ops_graph_create(const gims_op* ops /* HOST */, int32_t n_ops, void* stream, void** graph_exec_out);
int gims_ops_graph_launch(void* graph_exec, void* stream);
int gims_ops_graph_destroy(void* graph_exec);

/* ------------------------------------------------------------------------------------------------
 * Keypoint encoder front end: normalize_keypoints (gmatcher.py:26-33, with the reference's NHWC-as-NCHW
 * quirk resolved by the caller into cx, cy, scale) fused with the first Conv1d(2->c1)+BN(eval)+ReLU of
 * KeypointEncoder (gmatcher.py:87-97).  w1: [c1][2] and b1: [c1] have BatchNorm already folded in.
 * kpts: [n][2] pixel xy.  out: [n][c1].  Per-row normalisation parameters: norm[row_seg[i]] = {cx,cy,scale}.
 */
int gims_kenc_first(const float* kpts, const float* norm3 /* [n_seg][3] */, const int32_t* seg_of_row,
                    const float* w1, const float* b1, int32_t c1, float* out, int64_t n, void* stream);
/* Same without the ReLU (the use_layernorm=True variant normalises before activating). */
int gims_kenc_first_linear(const float* kpts, const float* norm3, const int32_t* seg_of_row,
                           const float* w1, const float* b1, int32_t c1, float* out, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm of the reference's use_layernorm=True variant + activation.
 * replaces: class LayerNorm (gmatcher.py:74-85) as MLP() inserts it between Conv1d and ReLU (gmatcher.py:19-23):
 *   y[r, :] = a2 * (x[r, :] - mean_r) / (std_r + eps) + b2   over the c channels of point r, std UNBIASED (c - 1),
 *   eps added to the std.  out (f32, may be NULL or == x) and/or out_hi/out_lo (SPL32 split planes, see gims_linear).
 * 2 <= c <= 512.
 */
int gims_layernorm_act(const float* x, int64_t ldx, int64_t rows, int32_t c, const float* a2, const float* b2, float eps,
                       int32_t act, float* out /* may be NULL */, int64_t ldo, uint16_t* out_hi /* may be NULL */,
                       uint16_t* out_lo, int64_t ld_split, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GraphSAGE mean aggregation over the adaptive graph (CSR by destination, both edge directions):
 *   out[i, :] = mean_{j in indices[indptr[i]:indptr[i+1]]} h[j, :]     (zero for isolated nodes)
 * replaces: the message passing of dgl.nn.SAGEConv(...,'mean') (call sites gmatcher.py:149-151,158).
 * Row ids in `indices` are absolute rows of h.  c must be a multiple of 4.
 */
int gims_sage_mean(const float* h, int64_t ldh, const int32_t* indptr, const int32_t* indices,
                   int32_t n, int32_t c, float* out, int64_t ldo, void* stream);
/* The same mean written directly as SPL32 split-bf16 planes (the A-operand layout of GIMS_PREC_BF16X3 pre-split GEMMs,
 * see gims_split_spl32): hi + lo of every f32 mean, no f32 copy.  ld_spl in bf16 elements, >= 2 * c (c rounded up to 32). */
int gims_sage_mean_split(const float* h, int64_t ldh, const int32_t* indptr, const int32_t* indices,
                         int32_t n, int32_t c, uint16_t* out_spl, int64_t ld_spl, void* stream);

/* Per-pair match statistics of a batch whose outputs are concatenated (the record the multi-GPU harness all-gathers,
 * SURVEY 8(e); the reference's eval loop keeps the same numbers per pair on the host, eval_homography.py:186-236):
 *   table[p] = {pair_id, n0, n1, offset of the pair's rows in matches0 / scores0}  (int32 x 4, device memory)
 *   out[p]   = {pair_id, n0, n1, #matches (matches0 >= 0), mean matching score over the matches (0 if none)}  (f32 x 5)
 * Fixed summation order: the result is deterministic. */
int gims_pair_stats(const int64_t* matches0, const float* scores0, const int32_t* table, int32_t n_pairs, float* out, void* stream);

/* Gather rows: dst[i, :] = src[idx[i], :]  (kept-keypoint compaction, gmatcher.py:244-249). */
int gims_gather_rows(const float* src, int64_t lds, const int32_t* idx, int32_t n, int32_t c,
                     float* dst, int64_t ldd, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Adaptive graph construction for a BATCH of images (models/agc.py:682-709, live subset), every stage one
 * launch for all images (blockIdx.y = image):
 *   cosine similarity (382-391) -> exact percentile threshold over the strict upper triangle (367-380,
 *   439-440) -> radius pairs in float64, inclusive (435-436) filtered by sim >= thr (445-447) ->
 *   connect_isolated_nodes (476-495) -> remove_small_components (497-516) -> fast_connect_components
 *   (518-565, one round) -> sorted relabel + bidirectional CSR (dgl.from_networkx, 704).
 * Per image: kpts [n][2] f32, desc [n][d] f32 (point-major, un-normalised); outputs (device): kept[n]
 * (first n_kept entries: sorted original ids), indptr[n+1] (first n_kept+1 valid), indices[max_edges_dir]
 * in kept-relabelled ids, info[8] = {n_kept, n_dir_edges, n_coarse_edges, n_iso_added,
 * n_components_after_removal, n_link_added, threshold bits (f32), flags: bit 0 = an edge / candidate buffer overflowed (repeat with a larger
 * max_edges_dir), bit 1 = see gims_agc_build_ex}.
 * `work` is scratch of at least gims_agc_workspace_bytes(images, n_images) bytes.
 * LIMIT: 2 <= n <= gims_agc_max_keypoints() = 32768 keypoints per image (GIMS_EINVAL above it; the reference -- NumPy / SciPy -- has no
 * limit and publishes runs with up to 21 163 kept keypoints, tools/files/rgbd1/record.txt:635).  What bounds it: a pair of node ids is one
 * packed 32-bit word (i << 16 | j), the sequential isolated-node walk keeps its ordered list in LDS (135 KB of 160 KB at 32768), and the
 * workspace reserves one word per pair of the strict upper triangle (n^2 * 2 bytes: 0.9 GB per image at 21 163, 2.1 GB at 32768) -- and
 * the ROBUST flow as much again for the half similarity matrix it stores (gims_agc_workspace_bytes_ex: 1.8 GB / 4.3 GB).  Images
 * above 16384 keypoints run the component search in global memory instead of LDS (same labels).
 * Exact-distance ties in the two sequential fix-ups resolve to the lowest node index.
 * Asynchronous; read info[] after synchronising the stream.
 */
typedef struct gims_agc_image {
  const float* kpts; const float* desc; int64_t ldd; int32_t n, d;
  int32_t* kept; int32_t* indptr; int32_t* indices; int32_t max_edges_dir; int32_t* info;
} gims_agc_image;

size_t gims_agc_workspace_bytes(const gims_agc_image* h_images /* HOST array */, int32_t n_images);   /* enough for either flow of gims_agc_build_ex */
/* Scratch bytes of gims_agc_build_ex(..., flags, ...): the default (window) flow never stores the N x N half similarity matrix, which is half of
 * the robust flow's workspace -- a batch of README-size images (15 k keypoints) asks for 0.5 GB per image instead of 0.9.  A call whose
 * images force the robust flow (d % 64 != 0 or d > 256; GIMS_AGC_ROBUST=1 in the environment) is sized for it whatever the flags say. */
size_t gims_agc_workspace_bytes_ex(const gims_agc_image* h_images /* HOST array */, int32_t n_images, int32_t flags);
int32_t gims_agc_max_keypoints(void);
int gims_agc_build(const gims_agc_image* h_images /* HOST array */, int32_t n_images, double radius, double percentile,
                   int32_t min_size, void* work, size_t work_bytes, void* stream);
/* The same with flags.  The percentile threshold is exact in both flows (the k-th smallest of the similarities as the library evaluates
 * them: float64 dot products of the normalised f32 rows, rounded once); they differ in how the entries that can decide it are found.
 * Default: a sample of one-pass half-precision similarities predicts a window of values that holds rank k, one pass over all N^2/2 of
 * them counts what lies below the window and lists what lies inside it, and only the listed entries (and the radius candidates) are
 * evaluated exactly.  The prediction is VERIFIED on the device against a rigorous bound of the half-precision error; if it did not hold,
 * bit 1 of info[7] is set (bit 0: edge capacity) and every other output of that image -- bit 0 included -- is to be discarded: repeat the call with
 * GIMS_AGC_ROBUST, which histograms every entry instead of predicting (about 0.2 ms more per 16 images of 4096).  Images of at most 1536
 * keypoints are "sampled" in full and never report bit 1. */
#define GIMS_AGC_ROBUST 1
int gims_agc_build_ex(const gims_agc_image* h_images /* HOST array */, int32_t n_images, double radius, double percentile,
                      int32_t min_size, int32_t flags, void* work, size_t work_bytes, void* stream);

/* Ingest a batch of images given in the reference's layout (descriptors channel-major (D,N), gmatcher.py:245) into
 * one row-concatenated point-major buffer: desc_out[row_off_i + n, :] = desc_i[:, n], same for keypoints and scores.
 * One launch for the whole batch (64x64 LDS-tile transpose). */
typedef struct gims_ingest_image {
  const float* kpts; const float* desc; int64_t ldd; const float* score; int32_t n, row_off;
} gims_ingest_image;

int gims_ingest_images(const gims_ingest_image* dev_images /* DEVICE array */, int32_t n_images, int32_t max_n, int32_t d,
                       float* desc_out, int64_t ldo, float* kpts_out, float* score_out /* may be NULL */, void* stream);

/* Pack the kept keypoints of a batch of images into the row-concatenated layout the rest of the path uses
 * (gmatcher.py:244-249): for image i with row offset ro_i and edge offset eo_i,
 *   feat[ro_i + r, :] = desc_i[kept_i[r], :],  kpts_out[ro_i + r] = kpts_i[kept_i[r]],  score_out likewise,
 *   seg[ro_i + r] = i,  indptr_out[ro_i + r] = indptr_i[r] + eo_i,  indices_out[eo_i + e] = indices_i[e] + ro_i
 * and indptr_out[n_rows_total] = n_edges_total. */
typedef struct gims_pack_image {
  const float* kpts; const float* desc; int64_t ldd; const float* score;
  const int32_t* kept; const int32_t* indptr; const int32_t* indices;
  int32_t n_kept, n_edges, row_off, edge_off;
} gims_pack_image;

int gims_pack_graphs(const gims_pack_image* dev_images /* DEVICE array */, int32_t n_images, int32_t max_kept,
                     int32_t max_edges, int32_t d, float* feat, int64_t ldf, float* kpts_out, float* score_out,
                     int32_t* seg, int32_t* indptr_out, int32_t* indices_out, int32_t n_rows_total,
                     int32_t n_edges_total, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Log-domain Sinkhorn optimal transport + mutual-argmax match selection.
 * replaces: log_optimal_transport / log_sinkhorn_iterations (gmatcher.py:41-69) and the selection block
 * (gmatcher.py:284-294).  The (N+1)x(M+1) couplings matrix is never materialised: the dustbin row and
 * column (= alpha) are handled analytically, and the iteration runs in the absorbed-potential form
 *   P_ij = exp(Z_ij + u_i + v_j);  u_i += log mu_i - log sum_j P_ij;  v_j += log nu_j - log sum_i P_ij
 * which is the reference recurrence u = log mu - LSE_j(Z + v), v = log nu - LSE_i(Z + u) rewritten so
 * that one sweep over Z serves both the row update and the following column update.
 * scores: f32 [n][ld] (inner N x M block, already divided by sqrt(D)).  Problems are independent
 * (one per image pair); `work` needs gims_sinkhorn_workspace_bytes(...) bytes.
 * Outputs per problem (device): matches0 [n] int64, matches1 [m] int64, mscores0 [n] f32, mscores1 [m]
 * f32, uv: u [n+1] then v [m+1] (log-potentials, so that OT = Z + u + v - norm can be rebuilt) then one
 * status word (0 = ok, 1 = a marginal left the finite range: matches are then all -1; 2 is transient: the on-chip
 * kernel could not get all its workgroups resident and gave up -- the call then re-solves THAT problem with a
 * dependency-free kernel before the selection runs (slow, rare), so a caller never sees 2 nor -1s caused by it).
 * Two implementations of the iteration loop (same recurrence, results agree to f32 rounding):
 *   streamed  -- one launch per iteration, Z read from HBM once per iteration (any size);
 *   resident  -- ONE launch for all iterations of a group of problems: exp(Z + u + v) is held in the registers and LDS
 *                of the 256 CUs and scaled lazily by cumulative row / column factors (re-derived from Z every 50 iterations and on the last),
 *                no HBM traffic inside the loop.  Used when the matrices fit on chip (n, m <= 4096, about 134 MB of
 *                matrix per launch) and are large enough to pay (>= 6 M entries); GIMS_OT_RESIDENT=0 / 2 in the
 *                environment forces streamed / resident.  gims_sinkhorn_plan() tells which one a call will take.
 *                Two on-chip kernels exist: the 2-D decomposition of csrc/sinkhorn2d.hip (default: a problem is cut into row
 *                groups, one per XCD, times 128-column blocks, one per CU; the row-sum exchange stays inside the XCD's L2, only
 *                a 0.5-KB column edge crosses XCDs; start potentials formed in the kernel) and the 1-D row-slab kernel of
 *                csrc/sinkhorn.hip (GIMS_OT_RES2=0; the cross-check).
 */
typedef struct gims_ot_problem {
  const float* scores; int64_t ld; int32_t n, m;
  int64_t* matches0; int64_t* matches1; float* mscores0; float* mscores1;
  float* uv;                                /* [n+1 + m+1 + 1] */
} gims_ot_problem;

size_t gims_sinkhorn_workspace_bytes(const gims_ot_problem* h_problems, int32_t n_problems);
/* 0: the call will run streamed; k > 0: resident, in k launches.  (Query only; no reference counterpart.) */
int gims_sinkhorn_plan(const gims_ot_problem* h_problems, int32_t n_problems, int32_t iters);
int gims_sinkhorn_match(const gims_ot_problem* h_problems /* HOST array */, int32_t n_problems, float alpha,
                        int32_t iters, float match_threshold, void* work, size_t work_bytes, void* stream);
/* The same two calls with flags.  GIMS_OT_STREAMED: never take an on-chip kernel -- they occupy every CU of the device for the
 * whole solve and wait on each other across workgroups, so a caller that runs several streams (or processes) on one GPU
 * concurrently must use the streamed kernels: next to another stream's kernels an on-chip solve cannot get its 256 workgroups
 * resident, gives up and is re-solved by the slow rescue path. */
#define GIMS_OT_STREAMED 1
int gims_sinkhorn_plan_ex(const gims_ot_problem* h_problems, int32_t n_problems, int32_t iters, int32_t flags);
int gims_sinkhorn_match_ex(const gims_ot_problem* h_problems /* HOST array */, int32_t n_problems, float alpha,
                           int32_t iters, float match_threshold, void* work, size_t work_bytes, int32_t flags, void* stream);

/* Diagnostics: how many on-chip solves gave up and were re-solved by a rescue path on the CURRENT device since the process started
 * (synchronises the device; -1 on error).  A given-up solve is re-solved inside the same call -- by the streamed kernels when a
 * give-up was seen during the last 256 calls (GIMS_OT_RESCUE=1: always), else by a one-workgroup kernel -- so results never
 * depend on it; the counter lets a caller (and the tests) see that it happened.  No reference counterpart. */
int64_t gims_sinkhorn_rescues(void);

/* ------------------------------------------------------------------------------------------------
 * Per-pair evaluation after the matcher -- batched over pairs, everything stays on the device.
 * replaces: the per-pair block of eval_homography.py:186-226 --
 *   torch_find_matches(kp0, kp1, H_gt, dist_thresh=3, n_iters=3) (utils/preprocess_utils.py:98-132, with warp_keypoints
 *   :86-96 and torch_cdist :74-78), precision / recall (:207-209, 222-226), cv2.getPerspectiveTransform on the four most
 *   confident matches (:216-217), cv2.findHomography(RANSAC) (:193, 218) and the corner error (:210, 219-223,
 *   utils/common.py:477-481).  The GT matching reproduces the reference's float32 arithmetic (index sets are exact); the
 *   two OpenCV calls are restated from their documented semantics -- the RANSAC is this build's own deterministic one,
 *   OpenCV's sampler is not reproducible.  Specification (float64; also restated in oracle/eval_oracle.py):
 *     hypothesis h = 0 .. iters-1: state = seed ^ (h * 0xD1342543DE82EF95); four DISTINCT indices into the ascending list of
 *     valid matches are successive splitmix64(state) % K values (duplicates skipped); H_h = exact 4-point homography
 *     (h22 = 1); score = #matches with ||H_h p0 - p1||^2 <= thresh^2; best = highest score, lowest h on ties; then ONE
 *     least-squares refit (normal equations of the 2K x 8 DLT system) on the inliers of the best hypothesis, and the final
 *     inlier mask / count / corner error under the refit model.  The four most confident matches of the DLT are taken
 *     with ties broken by the earlier match.
 * Outputs per pair: gt0 [n0] int32 (GT partner in image 1 or -1), inlier [n0] uint8 (1 where the match of keypoint i is
 * a RANSAC inlier), record float[16] (GIMS_EVAL_* below), homographies float[18] (H_dlt then H_ransac, row-major).
 */
typedef struct gims_eval_pair {
  const float* kpts0; const float* kpts1;   /* kept keypoints [n0][2], [n1][2] (what GMatcher returns as keypoints0/1) */
  const int64_t* matches0;                  /* [n0] */
  const float* mscores0;                    /* [n0] */
  int32_t n0, n1, height, width;            /* image 0 size for the corner error */
  float h_gt[9];                            /* ground-truth homography, row-major */
  int32_t* gt0; uint8_t* inlier; float* record; float* homographies;
} gims_eval_pair;
#define GIMS_EVAL_NVALID 0      /* matches0 > -1 */
#define GIMS_EVAL_NGT 1         /* GT correspondences found */
#define GIMS_EVAL_NCORRECT 2    /* matches0[i] == gt0[i] >= 0 */
#define GIMS_EVAL_NFN 3         /* unmatched keypoints that have a GT partner */
#define GIMS_EVAL_PRECISION 4
#define GIMS_EVAL_RECALL 5
#define GIMS_EVAL_NINLIERS 6
#define GIMS_EVAL_ERR_DLT 7     /* mean corner distance, -1 when no model */
#define GIMS_EVAL_ERR_RANSAC 8
#define GIMS_EVAL_DLT_OK 9
#define GIMS_EVAL_RANSAC_OK 10
size_t gims_eval_workspace_bytes(const gims_eval_pair* h_pairs, int32_t n_pairs, int32_t ransac_iters);
int gims_eval_pairs(const gims_eval_pair* h_pairs /* HOST array */, int32_t n_pairs, float dist_thresh, int32_t n_iters,
                    float ransac_thresh, int32_t ransac_iters, uint64_t seed, void* work, size_t work_bytes, void* stream);

/* Rebuild the full (n+1)x(m+1) OT matrix Z + u + v - norm (gmatcher.py:47,68) -- for forward_train and tests. */
int gims_ot_matrix(const float* scores, int64_t ld, int32_t n, int32_t m, float alpha, const float* uv,
                   float* out /* [(n+1)][(m+1)] */, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training loss of GMatcher.forward_train, forward value (gmatcher.py:333-386), from the solved potentials of
 * gims_sinkhorn_match -- the OT matrix is not materialised.  Per batch element b: scores / ld / n / m / uv as in
 * gims_ot_problem, kept0 / kept1 = the sorted ORIGINAL ids of the keypoints the adaptive graph kept (data['kept_kpts*_indices']).
 * gt: [n_gt][3] int64 rows (b, i0, i1) in ORIGINAL keypoint ids, -1 = no partner (train.py:113-125).  A row is remapped
 * through the kept lists (gmatcher.py:340-367); rows with a -1 or a dropped keypoint become (b, -1, -1), which the
 * reference's tensor indexing reads as the corner cell OT[N, M] (gmatcher.py:372), and count as negatives.  Each gathered
 * log-score is clamped to [-100, 0] and negated; positives and negatives are averaged per batch element (scatter_mean,
 * empty groups give 0), then over the batch, and weighted: out3 = {loss, pos_weight * pos, neg_weight * neg}.
 * loss_vec [n_gt] f32 and tag [n_gt + 2 * n_pairs] int32 are scratch (loss_vec holds the per-row losses afterwards, tag the row
 * classes followed by the per-(batch element, sign) group sizes: gims_train_loss_grad reads both).  Deterministic. */
typedef struct gims_loss_pair {
  const float* scores; int64_t ld; int32_t n, m; const float* uv; const int32_t* kept0; const int32_t* kept1;
} gims_loss_pair;
int gims_train_loss(const gims_loss_pair* dev_pairs /* DEVICE array */, int32_t n_pairs, const int64_t* gt, int32_t n_gt, float alpha,
                    float pos_weight, float neg_weight, float* loss_vec, int32_t* tag, float* out3, void* stream);

/* Backward of that loss through the Sinkhorn iterations (first differentiable kernel): d loss / d scores and
 * d loss / d bin_score, by reverse mode through the UNROLLED iterations of log_sinkhorn_iterations (gmatcher.py:41-47) -- what
 * autograd does in the reference.
 *   gims_sinkhorn_history: a forward solve with the streamed kernels that records the potentials after every iteration:
 *     h_hist[p] (HOST array of device pointers) receives (iters + 1) slots of n + m + 2 floats (u [n+1] then v [m+1]); slot 0 is
 *     unused (u_0 = v_0 = 0), slot k holds u_k, v_k.  problems[p].uv ends with the final potentials as in gims_sinkhorn_match.
 *   gims_train_loss_grad: scatters G = d loss / d out (out = Zc + u + v - norm at the gathered cells; zero where the clamp was
 *     active, gmatcher.py:374) into ZEROED per-problem buffers dz[p] [(n+1)][(m+1)]; `tag` is gims_train_loss's scratch output.
 *   gims_sinkhorn_backward: dz[p] holds G on entry and d loss / d Zc on exit (rows 0..n-1, columns 0..m-1 = d loss / d scores,
 *     pitch m + 1); dalpha[p] = the sum over the dustbin border = that problem's share of d loss / d bin_score. */
size_t gims_sinkhorn_history_floats(int32_t n, int32_t m, int32_t iters);
int gims_sinkhorn_history(const gims_ot_problem* h_problems, int32_t n_problems, float alpha, int32_t iters, float* const* h_hist /* HOST array */,
                          void* work /* gims_sinkhorn_workspace_bytes */, size_t work_bytes, void* stream);
size_t gims_sinkhorn_backward_workspace_bytes(const gims_ot_problem* h_problems, int32_t n_problems);
int gims_sinkhorn_backward(const gims_ot_problem* h_problems, int32_t n_problems, float alpha, int32_t iters, const float* const* h_hist /* HOST array */,
                           float* const* h_dz /* HOST array */, float* dalpha /* device [n_problems] */, void* work, size_t work_bytes, void* stream);
int gims_train_loss_grad(const gims_loss_pair* dev_pairs, int32_t n_pairs, const int64_t* gt, int32_t n_gt, float alpha, const int32_t* tag,
                         float pos_weight, float neg_weight, float* const* dev_dz_ptrs /* DEVICE array */, void* stream);

/* ------------------------------------------------------------------------------------------------
 * CAR-HyNet patch descriptor: the non-GEMM layers of carhynet/models.py:311-399.  Activations are NHWC
 * f32 ([patch][y][x][channel]); the 3x3 and 8x8 convolutions are gims_ch_im2col3 / a reshape + gims_linear (split-bf16x3),
 * the 1x1 convolutions gims_linear directly.  BatchNorm (eval) is folded into weights / biases by the caller.
 *   gims_ch_frn_stats: scale[p][c] = weight[c] * rsqrt(mean over the hw pixels of x^2 + eps)          (FRN, models.py:67-82)
 *   gims_ch_pool_hw:   ph[p][y][c] = mean_x(x*s+b), pw[p][x][c] = mean_y(x*s+b); s [p][c], b [c] may be NULL  (CoordAtt 141-143)
 *   gims_ch_gates:     CoordAtt's conv1(+BN folded)+h_swish over the h+w pooled rows, then conv_h / conv_w + sigmoid (144-151);
 *                      w1 [8][c], b1 [8], wh / ww [c][8], bh / bw [c];  h + w <= 64
 *   gims_ch_apply:     y = max((x*s[p][c] + b[c]) * ah[p][y][c] * aw[p][x][c], tau[c]); s/b, ah/aw, tau each optional (78-84,152,107)
 *   gims_ch_im2col3:   3x3 neighbourhoods (pad 1, stride 1|2) as SPL32 rows: column (ky*3+kx)*c + ch, zero-padded to kpad
 *   gims_ch_dwconv3:   depthwise 3x3 (pad 1) with wt [9][c], bias [c] (BatchNorm folded), optional ReLU6, optional
 *                      y += res_scale * res                                                            (172-180, 207, 220-233)
 *   gims_ch_l2norm:    y = x / sqrt(sum_c x^2 + eps) per row                                           (desc_l2norm, 9-21)
 *   gims_ch_relu6:     in-place clamp to [0, 6]
 */
int gims_ch_frn_stats(const float* x, int64_t patches, int32_t hw, int32_t c, const float* weight, float eps, float* scale, void* stream);
int gims_ch_pool_hw(const float* x, int64_t patches, int32_t h, int32_t w, int32_t c, const float* s, const float* b, float* ph, float* pw,
                    float* rowsq /* [p][y][c] sums of x^2 over x, may be NULL */, void* stream);
/* FRN scale from those row sums (one pass over the activation serves the statistics and both pools) */
int gims_ch_frn_from_rows(const float* rowsq, int64_t patches, int32_t h, int32_t w, int32_t c, const float* weight, float eps, float* scale, void* stream);
int gims_ch_gates(const float* ph, const float* pw, int64_t patches, int32_t h, int32_t w, int32_t c, const float* w1, const float* b1,
                  const float* wh, const float* bh, const float* ww, const float* bw,
                  const float* frn_scale /* [p][c] */, const float* frn_bias /* [c]; both NULL: the pools are used as they are */, float* ah, float* aw,
                  void* stream);
int gims_ch_apply(const float* x, int64_t patches, int32_t h, int32_t w, int32_t c, const float* s, const float* b, const float* ah, const float* aw,
                  const float* tau, float* y /* may be NULL */, uint16_t* y_split /* SPL32 pixel rows, may be NULL */, int64_t ld_split, void* stream);
int gims_ch_im2col3(const float* x, int64_t patches, int32_t h, int32_t w, int32_t c, int32_t stride, uint16_t* out, int64_t ld, int32_t kpad, void* stream);
int gims_ch_dwconv3(const float* x, int64_t patches, int32_t h, int32_t w, int32_t c, const float* wt, const float* bias, int32_t relu6_out,
                    const float* res, float res_scale, float* y /* may be NULL */, uint16_t* y_split /* may be NULL */, int64_t ld_split, void* stream);
/* SandGlass middle in one pass per pixel: z = ReLU6(w1 (w0 (x * a_w * a_h) + b0) + b1); w0 [16][c], w1 [c][16] (BatchNorm folded), c = 32 | 64 */
int gims_ch_gate_pw_pw(const float* x, int64_t patches, int32_t h, int32_t w, int32_t c, const float* ah, const float* aw, const float* w0,
                       const float* b0, const float* w1, const float* b1, float* z, void* stream);
/* Input stage (models.py:316-317): patches [n][32][32][3] f32 -> FRN(3) + TLU(3) -> the 3x3 neighbourhoods as SPL32 rows
 * [n*1024][pitch ld >= 128] for the first convolution's GEMM: column (ky*3+kx)*4 + c, K = 36 zero-padded to 64 (channel 3 is zero). */
int gims_ch_input_block(const float* patches, int64_t n, const float* frn_weight /* [>=3] */, const float* frn_bias, float eps, const float* tau,
                        uint16_t* out, int64_t ld, void* stream);
/* FRN (+ CoordAtt) + TLU of one layer in one pass over the activation (32x32x32, 16x16x64 or 8x8x128), one workgroup per patch with
 * the raw convolution output resident in LDS: y = max((x s + b) a_w a_h, tau), s = frn_weight * rsqrt(mean x^2 + eps) per (patch,
 * channel), gates as in gims_ch_gates (gate_w: HOST array of 6 device pointers w1 [8][c] (+BN folded), b1, wh [c][8], bh, ww, bw; NULL: no
 * CoordAtt).  Output f32 NHWC and / or SPL32 split-bf16 pixel rows. */
int gims_ch_frn_block(const float* x, int64_t patches, int32_t hw, int32_t c, const float* frn_weight, const float* frn_bias, float eps,
                      const float* const* gate_w, const float* tau, float* y, uint16_t* y_split, int64_t ld_split, void* stream);
/* x + SandGlass(x) (models.py:182-235 with the outer residual of 383-389: 2x + conv stack) for 32x32x32 or 16x16x64 activations,
 * one workgroup per patch with the activation resident in LDS; w: HOST array of 14 device pointers (BatchNorm folded):
 * dw0 [9][c], dw0 bias [c], CoordAtt w1 [8][c], b1 [8], wh [c][8], bh [c], ww [c][8], bw [c], pw0 [16][c], pw0 bias [16],
 * pw1 [c][16], pw1 bias [c], dw1 [9][c], dw1 bias [c].  Output: SPL32 split-bf16 pixel rows. */
int gims_ch_sandglass(const float* x, int64_t patches, int32_t hw, int32_t c, const float* const* w, uint16_t* out_split, int64_t ld_split, void* stream);
/* 3x3 convolution (pad 1, stride 1 | 2) + bias + FRN (+ CoordAtt) + TLU of one CAR-HyNet layer in ONE kernel, one workgroup per patch
 * (models.py:325-360: layer2 .. layer6).  x_split: the layer's input as SPL32 split-bf16 pixel rows [patches * hin * hin][pitch ldx >= 2 cin];
 * it is read once into LDS (zero border) and the convolution runs there as an implicit GEMM on the matrix cores (split-bf16x3: hi*hi +
 * hi*lo + lo*hi, like gims_linear); the raw output never leaves the chip: FRN statistic, CoordAtt gates (gate_w as in gims_ch_frn_block,
 * NULL: none) and TLU are applied to the accumulators and the result leaves as f32 NHWC (y) and / or SPL32 pixel rows (y_split).
 * w_packed: the weights [cout][cin][3][3] in MFMA fragment order, bf16: [step = (ky*3+kx) * cin/16 + ks][nb = cout/32][plane hi|lo][lane 64][8]
 * with lane = lh*32 + li holding out channel nb*32 + li, input channels 16 ks + 8 lh + (0..7) of tap (ky, kx).
 * Geometries: (hin, cin, cout, stride) = (32,32,32,1), (32,32,64,2), (16,64,64,1), (16,64,128,2), (8,128,128,1). */
int gims_ch_conv_block(const uint16_t* x_split, int64_t ldx, int64_t patches, int32_t hin, int32_t cin, int32_t cout, int32_t stride,
                       const uint16_t* w_packed, const float* bias, const float* frn_weight, const float* frn_bias, float eps,
                       const float* const* gate_w, const float* tau, float* y, uint16_t* y_split, int64_t ld_split, void* stream);
/* The FIRST layer the same way (models.py:316-323): patches [n][32][32][3] f32 -> FRN(3) + TLU(3) -> 3x3 convolution 3 -> 32 (weights packed like
 * above with the input channels zero-padded to 16) -> FRN(32) + CoordAtt + TLU, one workgroup per patch, nothing but the patch read and the result
 * written.  Replaces gims_ch_input_block + gims_linear + gims_ch_frn_block for this layer. */
int gims_ch_conv_block_first(const float* patches, int64_t n, const float* frn0_weight, const float* frn0_bias, float eps0, const float* tau0,
                             const uint16_t* w_packed, const float* bias, const float* frn_weight, const float* frn_bias, float eps,
                             const float* const* gate_w, const float* tau, float* y, uint16_t* y_split, int64_t ld_split, void* stream);
int gims_ch_l2norm(const float* x, int64_t rows, int32_t c, float eps, float* y, void* stream);
int gims_ch_relu6(float* x, int64_t total, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Patch extraction: the front-end stage between keypoint detection and CAR-HyNet,
 *   buildGaussianPyramid(img, 6, graydesc=False)  (utils/library.py:234-271)
 *   ComputePatches(k, pyramid, radius_size=64)    (utils/library.py:84-110)
 *   cv2.resize(p, (32, 32), INTER_AREA) / 255.0   (utils/common.py:884)
 * on uint8 HWC images, in OpenCV's uint8 fixed-point arithmetic as restated in gims_amd/csrc/patches.hip (PARITY UNPINNED
 * against OpenCV itself, which is not available here; bit-identical to oracle/patch_oracle.py).
 * gims_pyramid_layout: number of levels (nOctaves * 6), per-level {byte offset, h, w} inside ONE buffer of *pyr_bytes, and
 *   the scratch bytes gims_pyramid_build needs.  h_levels may be NULL to query the sizes only.
 * gims_pyramid_build: img [h][w][c] uint8 (device) -> pyr (device).  Level o*6+i: octave o, layer i; level 0 is the 2x
 *   INTER_LINEAR_EXACT upsampling of img, layer 0 of octave o > 0 the INTER_NEAREST halving of level (o-1)*6+3, every other
 *   level the uint8 GaussianBlur of its predecessor.
 * gims_patch_extract: kp4 [n][4] f32 = (x, y, size, angle) and kp_octave [n] int32 = cv2.KeyPoint.octave (packed octave /
 *   layer, utils/library.py:16-35), all on the device.  out [n][32][32][3] f32 in [0, 1] (NHWC: the input layout of the
 *   CAR-HyNet kernels).  *bad_count (device int32) = keypoints whose octave / layer lie outside the pyramid (zero patches;
 *   the reference would raise IndexError).  c must be 3. */
typedef struct gims_pyr_level { int64_t offset; int32_t h, w; } gims_pyr_level;
int gims_pyramid_layout(int32_t h, int32_t w, int32_t c, gims_pyr_level* h_levels /* HOST, may be NULL */, int32_t capacity, int32_t* n_levels,
                        size_t* pyr_bytes, size_t* scratch_bytes);
int gims_pyramid_build(const uint8_t* img, int32_t h, int32_t w, int32_t c, uint8_t* pyr, void* scratch, void* stream);
int gims_patch_extract(const uint8_t* pyr, const gims_pyr_level* dev_levels, int32_t n_levels, const float* kp4, const int32_t* kp_octave,
                       int32_t n_kp, float* out, int32_t* bad_count, void* stream);
/* The per-keypoint arithmetic of gims_patch_extract on its own: A_out [n][6] f64 = the 2x3 map ComputePatches hands to
 * cv2.warpAffine (utils/library.py:96-106, row-major a00 a01 a02 a10 a11 a12), level_out [n] = the pyramid level it warps,
 * (octave - firstOctave) * 6 + layer (utils/library.py:100; unchecked).  The same device function as inside patch extraction;
 * it exists so that this arithmetic can be pinned against the reference's own (tests/golden/patch_affine_*.npz). */
int gims_patch_affine(const float* kp4, const int32_t* kp_octave, int32_t n_kp, double* A_out, int32_t* level_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training step (SURVEY 8f, row f3 = a25): GMatcher.forward(data, mode='train') on a module in train() mode followed by
 * loss.backward() (models/gmatcher.py:309-386 under train.py:100, 136-137).  gims_amd/trainstep.py composes the entry points
 * below (with gims_agc_build, gims_sage_mean, gims_sinkhorn_match, gims_train_loss, gims_sinkhorn_history,
 * gims_train_loss_grad and gims_sinkhorn_backward) into the forward and the reverse pass behind a torch.autograd.Function
 * whose inputs are the module's parameters; activations are row-major f32 [keypoint rows][channels].
 *
 * gims_gemm_f32: for z < batch   C_z = alpha * op(A_z) op(B_z)^T + beta * C_z (+ bias[n]) (+ residual_z), then act.
 *   op(A)(m, k) = ta ? A[k * lda + m] : A[m * lda + k];  op(B)(n, k) = tb ? B[k * ldb + n] : B[n * ldb + k].
 *   X_z = X + z * s{a,b,c,r} elements.  Arithmetic: f32 operands split on the fly into two (three) bf16 parts, three (six)
 *   bf16 MFMA passes per product, f32 accumulation (`precision`).  Replaces every torch matmul / conv1d(k=1) / einsum of the step and of
 *   its autograd: nn.Conv1d forward, grad_input, grad_weight (gmatcher.py:11-24, 99-125, 202-205), attention (35-39), the
 *   score einsum (273-275).  Any m, n, k >= 0; 16-byte aligned operands with pitches that are multiples of 4 take the
 *   vector path, anything else is read element by element.  `flags` and `splits` are set by the library. */
typedef struct gims_gemm {
  const float* a; const float* b; float* c;
  const float* bias;          /* [n] or NULL */
  const float* residual;      /* [m][ldr] or NULL */
  int64_t lda, ldb, ldc, ldr;
  int64_t sa, sb, sc, sr;
  int32_t m, n, k, batch;
  int32_t ta, tb;
  int32_t act;                /* GIMS_ACT_* */
  int32_t flags;
  float alpha, beta;
  float* work;                /* optional split-K workspace (device), work_floats floats: used when the output has too few tiles */
  int64_t work_floats;        /* to fill the chip and k >= 512; partial sums are added in a fixed order.  NULL: never split */
  int32_t splits;             /* set by the library */
  int32_t precision;          /* GIMS_PREC_BF16X3 (16 mantissa bits per operand) or GIMS_PREC_BF16X6 (24 bits: the f32 class) */
} gims_gemm;
int gims_gemm_f32(const gims_gemm* g, void* stream);

/* Row segments of a batch: one segment per call of the module in the reference (all image-0 rows, all image-1 rows). */
typedef struct gims_segments { int32_t n; int32_t off[8]; int32_t rows[8]; } gims_segments;
/* nn.BatchNorm1d in train() mode after a Conv1d (gmatcher.py:17-22), optionally with the ReLU that follows it:
 * statistics per (segment, channel) over the segment's rows, biased variance for the normalisation; running_mean /
 * running_var (may be NULL) are updated segment by segment with `momentum` and the unbiased variance, exactly the sequence
 * of updates the reference's two calls per layer make.  save [segments][c][2] = (mean, 1/sqrt(var + eps)) for the backward
 * pass; work: gims_batchnorm_workspace_floats floats.  Backward: dy is the gradient of the (post-ReLU) output; the ReLU mask
 * is recomputed from x; dgamma / dbeta are the sums over all segments. */
size_t gims_batchnorm_workspace_floats(const gims_segments* sg, int32_t c);
int gims_batchnorm_train_forward(const float* x, int64_t ld, int32_t c, const gims_segments* sg, const float* gamma, const float* beta, float eps,
                                 float momentum, float* running_mean, float* running_var, float* save, float* y, int64_t ldy, int32_t relu,
                                 float* work, void* stream);
int gims_batchnorm_train_backward(const float* x, int64_t ld, const float* dy, int64_t ldd, int32_t c, const gims_segments* sg, const float* save,
                                  const float* gamma, const float* beta, int32_t relu, float* dx, int64_t ldx, float* dgamma, float* dbeta,
                                  float* work, void* stream);
/* Reverse pass of gims_layernorm_act (the reference's LayerNorm, gmatcher.py:74-85, + ReLU): dy = gradient of the (post-ReLU) output, the
 * mask is recomputed from x.  dx [rows][ldo]; g_bias / g_scale [rows][c] receive the masked dy and dy * xhat, whose column sums
 * (gims_colsum) are the gradients of b_2 and a_2. */
int gims_layernorm_backward(const float* x, int64_t ldx, const float* dy, int64_t ldd, int64_t rows, int32_t c, const float* a2, const float* b2,
                            float eps, int32_t relu, float* dx, int64_t ldo, float* g_bias, float* g_scale, void* stream);
/* Attention of the training step for all images and heads of one GNN layer, without the probability matrices (gmatcher.py:35-39 inside
 * forward_train, :309-386; replaces, per image: the scores product, gims_softmax_rows and the product with V, and in the reverse pass four products
 * and gims_softmax_rows_backward).  qkv [rows][ld]: the packed projections Q | K | V (each d = 64 * heads columns, head h in columns 64 h ..
 * 64 h + 63 of its third: gims_head_pack's layout).  problems: a HOST array; problem i attends the query rows [q_off, q_off + nq) to the source
 * rows [k_off, k_off + nk) (self: the same image, cross: the other one).
 * forward: o [rows][ldo] = softmax(scale * Q K^T) V and lse [heads][rows] = max + log(sum) of every score row (what the reverse pass needs
 * instead of P).  backward: d_qkv [rows][lddq] from d_o, o, lse (every row must be a query row of exactly one problem and a source row of exactly
 * one for d_qkv to be written completely).  The forward is exact f32 on the matrix cores (v_mfma_f32_32x32x2_f32); the reverse pass -- linear in
 * its operands -- as reverse_precision says.  Fixed summation orders.  work:
 * gims_train_attention_workspace_floats(rows, heads) floats, 16-byte aligned. */
typedef struct gims_train_attn_problem {
  int32_t q_off, nq, k_off, nk;
} gims_train_attn_problem;
typedef struct gims_train_attn_args {
  const float* qkv;
  int64_t ld, rows;
  int32_t d, heads;
  float scale;                               /* 1 / sqrt(64) */
  int32_t n_problems;
  const gims_train_attn_problem* problems;   /* host memory */
  float* o;
  int64_t ldo;
  float* lse;
  const float* d_o;                          /* backward only */
  int64_t lddo;
  float* d_qkv;                              /* backward only */
  int64_t lddq;
  float* work;
  size_t work_floats;
  int32_t reverse_precision;                 /* backward only: GIMS_TRAIN_ATTN_REVERSE_* */
} gims_train_attn_args;
#define GIMS_TRAIN_ATTN_REVERSE_F32 0     /* exact f32 products (v_mfma_f32_32x32x2_f32), like the forward */
#define GIMS_TRAIN_ATTN_REVERSE_BF16X3 1  /* operands split into two bf16 parts, three passes (v_mfma_f32_32x32x16_bf16): 16-bit-mantissa products */
size_t gims_train_attention_workspace_floats(int64_t rows, int32_t heads);
int gims_train_attention_forward(const gims_train_attn_args* args, void* stream);
int gims_train_attention_backward(const gims_train_attn_args* args, void* stream);
/* softmax over the last dimension of `batch` matrices [rows][cols] (pitch ld, matrix stride `stride`), in place
 * (gmatcher.py:37), and its backward: dp <- prob * (dp - rowsum(dp * prob)), in place on dp. */
int gims_softmax_rows(float* s, int64_t ld, int64_t rows, int32_t cols, int32_t batch, int64_t stride, void* stream);
int gims_softmax_rows_backward(const float* prob, float* dp, int64_t ld, int64_t rows, int32_t cols, int32_t batch, int64_t stride, void* stream);
/* out[ch] = beta * out[ch] + sum over rows of x[row][ch] (bias gradients), fixed summation order.  work: zero-initialised
 * once by the caller (gims_colsum_workspace_floats floats), left zeroed where it must be. */
size_t gims_colsum_workspace_floats(int64_t rows, int32_t c);
int gims_colsum(const float* x, int64_t ld, int64_t rows, int32_t c, float beta, float* out, float* work, void* stream);
#define GIMS_EW_SCALE 0      /* out = alpha * a */
#define GIMS_EW_ADD 1        /* out = a + alpha * b */
#define GIMS_EW_RELU_MASK 2  /* out = b > 0 ? a : 0   (gradient of ReLU, b = the ReLU's output) */
#define GIMS_EW_RELU 3       /* out = max(a, 0) */
#define GIMS_EW_ACC 4        /* out += alpha * a */
int gims_elementwise(int32_t op, float* out, int64_t ldo, const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int32_t cols,
                     float alpha, void* stream);
/* dst[i0*d0 + i1*d1 + i2*d2] (= | +=) src[i0*s0 + i1*s1 + i2*s2] over [n0][n1][n2]: the reference's interleaved heads
 * (channel = d * heads + h, gmatcher.py:108-113) <-> contiguous heads, for weights on the way in and gradients on the way out. */
int gims_permute3(float* dst, const float* src, int32_t n0, int32_t n1, int32_t n2, int64_t d0, int64_t d1, int64_t d2, int64_t s0, int64_t s1,
                  int64_t s2, int32_t accumulate, void* stream);
/* Every head-interleave permutation of one attention layer in one launch.  proj_w / proj_b: HOST arrays of 3 device pointers (the q, k, v
 * projections in the reference's layout, [d][d] and [d]); merge_w [d][d].  to_params = 0: pack them into wqkv [3d][d], bqkv [3d] (rows grouped by
 * head) and wm [d][d] (columns grouped by head); to_params = 1: the reverse map (weight gradients back to the parameters' layout). */
int gims_head_pack(float* const* proj_w, float* const* proj_b, float* merge_w, float* wqkv, float* bqkv, float* wm, int32_t d, int32_t heads,
                   int32_t to_params, void* stream);
/* gradient of gims_sage_mean with respect to its input: out_j = sum over neighbours i of j of g_i / deg_i (symmetric CSR). */
int gims_sage_mean_transposed(const float* g, int64_t ldg, const int32_t* indptr, const int32_t* indices, int32_t n, int32_t c, float* out,
                              int64_t ldo, void* stream);
/* normalize_keypoints (gmatcher.py:26-33) as a tensor: out [n][2] = (kpts - norm3[seg][0..1]) / norm3[seg][2]. */
int gims_normalize_keypoints(const float* kpts, const float* norm3, const int32_t* seg_of_row, int64_t n, float* out, void* stream);

/* ---- optimizer step of the training loop (train.py:52-57 builds torch.optim.Adam over three parameter groups -- BatchNorm weights +
 * bin_score / other weights with weight decay / biases --, train.py:138 calls optimizer.step()) ----
 * One fused multi-tensor Adam step over HOST tables of device pointers (contiguous float32 tensors): ~count/80 launches for the
 * whole model instead of one list-kernel per operation and 64-tensor bucket.  Arithmetic = torch's _single_tensor_adam (amsgrad =
 * False, maximize = False), operation by operation in float32; the step-dependent scalars (lr / (1 - beta1^step), sqrt(1 - beta2^step))
 * are formed in double and rounded once.  `step` is the 1-based count AFTER this step's increment, per group (torch keeps it per
 * parameter; all parameters of a group that are stepped together share it).  Tensors with n = 0 are skipped. */
typedef struct gims_adam_tensor { float* param; const float* grad; float* exp_avg; float* exp_avg_sq; int64_t n; int32_t group; int32_t reserved; } gims_adam_tensor;
typedef struct gims_adam_group { double lr, beta1, beta2, eps, weight_decay; int64_t step; } gims_adam_group;
int gims_adam_step(const gims_adam_tensor* tensors, int32_t count, const gims_adam_group* groups, int32_t n_groups /* <= 8 */, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GIMS_HIP_H */

// Host-side AddressSanitizer harness ("compile-time -fsanitize=address for host C++").  CPU container
// only -- never on the GPU pool.  The library's HOST code (argument validation, workspace / plan / layout arithmetic, table
// builders, error strings, the replay loop) is compiled with -fsanitize=address (device code untouched: -fno-gpu-sanitize)
// and driven through every entry point that can run without a GPU: calls either return before touching HIP (GIMS_EINVAL,
// size queries) or fail cleanly inside the runtime (GIMS_EHIP: no device here).  ASan aborts on any heap / stack / global
// overflow, use-after-free or leak it sees on the way.    Build + run: tools/asan/run.sh
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../include/gims_hip.h"

static int checks = 0;
#define EXPECT(cond) do { ++checks; if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s  (last error: %s)\n", __FILE__, __LINE__, #cond, gims_last_error()); exit(1); } } while (0)

int main() {
  EXPECT(gims_abi_version() == GIMS_ABI_VERSION);
  // ---- Sinkhorn: size / plan arithmetic over ragged problem lists (host vectors, per-problem loops)
  std::vector<gims_ot_problem> pr;
  for (int i = 0; i < 40; ++i) {
    gims_ot_problem q; memset(&q, 0, sizeof(q));
    q.n = 600 + 7 * i; q.m = 640 - 5 * i; q.ld = (q.m + 3) / 4 * 4;
    pr.push_back(q);
  }
  EXPECT(gims_sinkhorn_workspace_bytes(pr.data(), (int)pr.size()) > 0);
  EXPECT(gims_sinkhorn_workspace_bytes(nullptr, 3) == 0);
  EXPECT(gims_sinkhorn_plan(pr.data(), (int)pr.size(), 100) >= 0);
  for (int n : {1, 31, 1024, 4096, 5000, 16384}) {
    gims_ot_problem q; memset(&q, 0, sizeof(q)); q.n = n; q.m = n; q.ld = (n + 3) / 4 * 4;
    EXPECT(gims_sinkhorn_workspace_bytes(&q, 1) > 0);
    EXPECT(gims_sinkhorn_plan(&q, 1, 100) >= 0);
  }
  EXPECT(gims_sinkhorn_match(nullptr, 1, 1.f, 10, 0.2f, nullptr, 0, nullptr) == GIMS_EINVAL);
  EXPECT(gims_sinkhorn_match(pr.data(), (int)pr.size(), 1.f, -1, 0.2f, (void*)0x1000, 1 << 20, nullptr) == GIMS_EINVAL);
  EXPECT(strlen(gims_last_error()) > 0);
  // ---- adaptive graph: workspace arithmetic, validation
  std::vector<gims_agc_image> im(5);
  for (size_t i = 0; i < im.size(); ++i) { memset(&im[i], 0, sizeof(im[i])); im[i].n = 100 + 900 * (int)i; im[i].d = 256; im[i].ldd = 256; im[i].max_edges_dir = 64 * im[i].n; }
  EXPECT(gims_agc_workspace_bytes(im.data(), (int)im.size()) > 0);
  EXPECT(gims_agc_build(nullptr, 0, 15.0, 2.0, 7, nullptr, 0, nullptr) != GIMS_OK);
  {   // rounds 4-5: gims_agc_build_ex (flags), the keypoint limit by name, workspace arithmetic at the largest published size and at the limit
    EXPECT(gims_agc_max_keypoints() == 32768);
    EXPECT(gims_agc_build_ex(nullptr, 0, 15.0, 2.0, 7, GIMS_AGC_ROBUST, nullptr, 0, nullptr) != GIMS_OK);
    gims_agc_image big; memset(&big, 0, sizeof(big));
    big.d = 256; big.ldd = 256;
    for (int n : {2, 16384, 16385, 21163, 32768}) {
      big.n = n; big.max_edges_dir = 64 * n;
      const size_t wb = gims_agc_workspace_bytes(&big, 1);
      EXPECT(wb > (size_t)n * (size_t)(n - 1) * 2);                                   // one word per pair of the strict upper triangle is in there
    }
    big.n = 32769; big.max_edges_dir = 64;                                             // over the limit: refused with the limit in the message
    big.kpts = (const float*)0x1000; big.desc = (const float*)0x2000; big.kept = (int32_t*)0x3000; big.indptr = (int32_t*)0x4000;
    big.indices = (int32_t*)0x5000; big.info = (int32_t*)0x6000;
    EXPECT(gims_agc_build_ex(&big, 1, 15.0, 2.0, 7, 0, (void*)0x7000, (size_t)1 << 40, nullptr) == GIMS_EINVAL);
    EXPECT(strstr(gims_last_error(), "32768") != nullptr);
    big.n = 100; big.d = 48;                                                           // descriptor width not a multiple of 32
    EXPECT(gims_agc_build_ex(&big, 1, 15.0, 2.0, 7, 0, (void*)0x7000, (size_t)1 << 40, nullptr) == GIMS_EINVAL);
    big.d = 256;                                                                       // a workspace that is too small
    EXPECT(gims_agc_build_ex(&big, 1, 15.0, 2.0, 7, GIMS_AGC_ROBUST, (void*)0x7000, 64, nullptr) == GIMS_EINVAL);
    // round 6: per-flow workspace sizes -- the default flow does not hold the half N x N matrix; a descriptor width that forces the robust flow
    // is sized for it whatever the flags say; a workspace sized for the default flow is refused by a robust build
    for (int n : {1000, 21163}) {
      big.n = n; big.d = 256; big.max_edges_dir = 64 * n;
      const size_t w0 = gims_agc_workspace_bytes_ex(&big, 1, 0), w1 = gims_agc_workspace_bytes_ex(&big, 1, GIMS_AGC_ROBUST);
      EXPECT(w1 == gims_agc_workspace_bytes(&big, 1) && w1 >= w0 + (size_t)n * (size_t)n * 2 && w0 > (size_t)n * (size_t)(n - 1) * 2);
    }
    big.n = 1000; big.d = 96; big.max_edges_dir = 64000;
    EXPECT(gims_agc_workspace_bytes_ex(&big, 1, 0) == gims_agc_workspace_bytes_ex(&big, 1, GIMS_AGC_ROBUST));
    big.d = 256;
    EXPECT(gims_agc_build_ex(&big, 1, 15.0, 2.0, 7, GIMS_AGC_ROBUST, (void*)0x7000, gims_agc_workspace_bytes_ex(&big, 1, 0), nullptr) == GIMS_EINVAL);
    EXPECT(gims_agc_workspace_bytes_ex(nullptr, 0, 0) == 0);
  }
  {   // round 5: guarded launches (gims_attn_guard) -- validation only
    gims_attn_args aa; memset(&aa, 0, sizeof(aa));
    EXPECT(gims_attention_ex(nullptr, nullptr) == GIMS_EINVAL);
    EXPECT(gims_attention_ex(&aa, nullptr) == GIMS_EINVAL);
    aa.qkv = (const uint16_t*)0x1000; aa.ld = 1536; aa.k_col = 256; aa.v_col = 512; aa.problems = (const gims_attn_problem*)0x2000; aa.n_problems = 1;
    aa.max_n_q = 64; aa.n_heads = 4; aa.out = (float*)0x3000; aa.ld_out = 256;
    aa.guard.stat = (uint64_t*)0x4000; aa.guard.kind = GIMS_GUARD_PEAKED; aa.guard.n_heads = 4;
    EXPECT(gims_attention_ex(&aa, nullptr) == GIMS_EINVAL);                            // a guard without GIMS_ATTN_X3
    aa.flags = GIMS_ATTN_X3; aa.guard.kind = 7;
    EXPECT(gims_attention_ex(&aa, nullptr) == GIMS_EINVAL);                            // unknown guard kind
    aa.guard.kind = GIMS_GUARD_RANGE; aa.guard.n_heads = 16;
    EXPECT(gims_attention_ex(&aa, nullptr) == GIMS_EINVAL);                            // more heads than one wave evaluates
    gims_linear_args lg; memset(&lg, 0, sizeof(lg));
    lg.a0 = (const float*)0x1000; lg.w = (const void*)0x2000; lg.out_f32 = (float*)0x3000; lg.m = lg.n = lg.k = lg.k0 = 64; lg.lda0 = lg.ldw = lg.ldc = 64;
    lg.guard.stat = (uint64_t*)0x4000; lg.guard.kind = GIMS_GUARD_PEAKED; lg.guard.n_heads = 4;
    EXPECT(gims_linear(&lg, nullptr) == GIMS_EINVAL);                                  // a guard on a launch without pre-split operands
  }
  {   // round 5: training attention (gims_train_attention_*) -- argument validation and workspace arithmetic (no launch)
    EXPECT(gims_train_attention_workspace_floats(0, 4) == 0);
    EXPECT(gims_train_attention_workspace_floats(4096, 4) >= (size_t)4096 * 4 * (1 + 3 * 64));
    EXPECT(gims_train_attention_workspace_floats(3, 3) % 4 == 0 || gims_train_attention_workspace_floats(3, 3) > 9);      // D block rounded to 16 bytes
    std::vector<gims_train_attn_problem> tp(40);
    for (int i = 0; i < 40; ++i) { tp[i].q_off = 100 * i; tp[i].nq = 100; tp[i].k_off = 100 * (i ^ 1); tp[i].nk = 100; }
    gims_train_attn_args ta; memset(&ta, 0, sizeof(ta));
    EXPECT(gims_train_attention_forward(nullptr, nullptr) == GIMS_EINVAL);
    EXPECT(gims_train_attention_forward(&ta, nullptr) == GIMS_EINVAL);
    ta.qkv = (const float*)0x10000; ta.ld = 768; ta.rows = 4000; ta.d = 256; ta.heads = 4; ta.scale = 0.125f; ta.n_problems = 40; ta.problems = tp.data();
    ta.o = (float*)0x20000; ta.ldo = 256; ta.lse = (float*)0x30000;
    ta.heads = 8;
    EXPECT(gims_train_attention_forward(&ta, nullptr) == GIMS_EINVAL);                 // head dimension 32
    ta.heads = 4; ta.ld = 767;
    EXPECT(gims_train_attention_forward(&ta, nullptr) == GIMS_EINVAL);                 // pitch
    ta.ld = 768; tp[38].nk = 101;
    EXPECT(gims_train_attention_forward(&ta, nullptr) == GIMS_EINVAL);                 // sources beyond the rows
    EXPECT(strstr(gims_last_error(), "problem 38") != nullptr);
    tp[38].nk = 100; tp[7].nq = 0;
    EXPECT(gims_train_attention_forward(&ta, nullptr) == GIMS_EINVAL);                 // an empty problem
    tp[7].nq = 100;
    EXPECT(gims_train_attention_forward(&ta, nullptr) == GIMS_EINVAL);                 // several key ranges and no workspace
    ta.work = (float*)0x40004;
    EXPECT(gims_train_attention_forward(&ta, nullptr) == GIMS_EINVAL);                 // misaligned workspace
    ta.work = (float*)0x40000; ta.work_floats = 16;
    EXPECT(gims_train_attention_forward(&ta, nullptr) == GIMS_EINVAL);                 // too small
    ta.work_floats = gims_train_attention_workspace_floats(ta.rows, ta.heads);
    EXPECT(gims_train_attention_backward(&ta, nullptr) == GIMS_EINVAL);                // no gradient tensors
    ta.d_o = (const float*)0x50000; ta.lddo = 256; ta.d_qkv = (float*)0x60000; ta.lddq = 768; ta.reverse_precision = 5;
    EXPECT(gims_train_attention_backward(&ta, nullptr) == GIMS_EINVAL);                // unknown reverse precision
  }
  // ---- evaluation: workspace arithmetic
  std::vector<gims_eval_pair> ep(3);
  for (size_t i = 0; i < ep.size(); ++i) { memset(&ep[i], 0, sizeof(ep[i])); ep[i].n0 = 500 + (int)i; ep[i].n1 = 400; ep[i].height = 480; ep[i].width = 640; }
  EXPECT(gims_eval_workspace_bytes(ep.data(), (int)ep.size(), 2000) > 0);
  EXPECT(gims_eval_pairs(nullptr, 0, 3.f, 3, 3.f, 100, 1, nullptr, 0, nullptr) != GIMS_OK);
  // ---- pyramid layout (host loops over octaves / levels, caller-provided table with exact and short capacity)
  int32_t nl = 0; size_t pb = 0, sb = 0;
  EXPECT(gims_pyramid_layout(480, 640, 3, nullptr, 0, &nl, &pb, &sb) == GIMS_OK && nl > 0 && nl % 6 == 0 && pb > 0 && sb > 0);
  std::vector<gims_pyr_level> lv(nl);
  EXPECT(gims_pyramid_layout(480, 640, 3, lv.data(), nl, &nl, &pb, &sb) == GIMS_OK);
  EXPECT(lv[0].h == 960 && lv[0].w == 1280 && lv[6].h == 480 && (size_t)lv[nl - 1].offset < pb);
  EXPECT(gims_pyramid_layout(480, 640, 3, lv.data(), nl - 1, &nl, &pb, &sb) == GIMS_EINVAL);
  EXPECT(gims_pyramid_layout(0, 640, 3, nullptr, 0, &nl, &pb, &sb) == GIMS_EINVAL);
  for (int h : {1, 2, 3, 7, 33, 75, 1201}) {          // tiny images have no octave at all: a clean GIMS_EINVAL
    const int rc = gims_pyramid_layout(h, 2 * h + 1, 3, nullptr, 0, &nl, &pb, &sb);
    EXPECT(rc == GIMS_OK || (h < 3 && rc == GIMS_EINVAL));
  }
  EXPECT(gims_pyramid_build(nullptr, 10, 10, 3, nullptr, nullptr, nullptr) == GIMS_EINVAL);
  EXPECT(gims_patch_extract(nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr) == GIMS_EINVAL);
  // ---- replay loop: empty table, unknown op kind, a linear op that fails validation
  EXPECT(gims_run_ops(nullptr, 0, nullptr) == GIMS_EINVAL);
  std::vector<gims_op> ops(3);
  memset(ops.data(), 0, sizeof(gims_op) * ops.size());
  EXPECT(gims_run_ops(ops.data(), 0, nullptr) == GIMS_OK);
  ops[0].kind = 7;
  EXPECT(gims_run_ops(ops.data(), 1, nullptr) == GIMS_EINVAL);
  ops[0].kind = GIMS_OP_LINEAR;                       // all-null linear arguments
  EXPECT(gims_run_ops(ops.data(), 3, nullptr) != GIMS_OK);
  ops[0].kind = GIMS_OP_ATTENTION;
  EXPECT(gims_run_ops(ops.data(), 3, nullptr) != GIMS_OK);
  EXPECT(gims_run_ops_timed(ops.data(), 1, nullptr, nullptr) == GIMS_EINVAL);
  // ---- validation of the remaining entry points (every call returns before touching memory it was not given)
  gims_linear_args la; memset(&la, 0, sizeof(la));
  EXPECT(gims_linear(&la, nullptr) != GIMS_OK);
  EXPECT(gims_linear(nullptr, nullptr) != GIMS_OK);
  EXPECT(gims_linear_batch(nullptr, 0, 0, 0, GIMS_PREC_F32, nullptr) != GIMS_OK);
  EXPECT(gims_attention(nullptr, 0, 0, 0, 0, nullptr, 0, 0, 0, nullptr, 0, nullptr, nullptr, 0, 0, nullptr) == GIMS_EINVAL);
  EXPECT(gims_attention_stat(nullptr, 0, 0, 0, 0, nullptr, 0, 0, 0, nullptr, 0, nullptr, nullptr, 0, 0, nullptr, nullptr) == GIMS_EINVAL);
  EXPECT(gims_attention_stat((const uint16_t*)0x1000, 768, 0, 256, 512, (const gims_attn_problem*)0x2000, 1, 64, 4, (float*)0x3000, 256, nullptr, nullptr, 0, 0,
                             (uint64_t*)0x4004, nullptr) == GIMS_EINVAL);      // misaligned statistics accumulator
  {   // round 6: the launch counters (host-side state only)
    uint64_t cnt[8] = {9, 9, 9, 9, 9, 9, 9, 9};
    EXPECT(gims_attention_launch_counts(cnt, 8, 1) == GIMS_OK && cnt[GIMS_ATTN_KERNEL_KINDS] == 0 && cnt[7] == 0);
    EXPECT(gims_attention_launch_counts(cnt, 3, 0) == GIMS_OK && cnt[0] == 0 && cnt[2] == 0);
    EXPECT(gims_attention_launch_counts(nullptr, 4, 0) == GIMS_EINVAL && gims_attention_launch_counts(nullptr, 0, 0) == GIMS_OK);
    gims_attn_args aa; memset(&aa, 0, sizeof(aa));                                    // a guard without GIMS_ATTN_X3 is refused before anything is launched
    aa.qkv = (const uint16_t*)0x1000; aa.ld = 1536; aa.k_col = 256; aa.v_col = 512; aa.problems = (const gims_attn_problem*)0x2000; aa.n_problems = 1;
    aa.max_n_q = 64; aa.n_heads = 4; aa.out = (float*)0x3000; aa.ld_out = 256; aa.guard.stat = (uint64_t*)0x4000; aa.guard.kind = GIMS_GUARD_PEAKED;
    aa.guard.n_heads = 4; aa.guard.max_thr = 0.5;
    EXPECT(gims_attention_ex(&aa, nullptr) == GIMS_EINVAL);
  }
  EXPECT(gims_patch_affine(nullptr, nullptr, 3, nullptr, nullptr, nullptr) == GIMS_EINVAL);
  EXPECT(gims_patch_affine(nullptr, nullptr, 0, nullptr, nullptr, nullptr) == GIMS_OK);          // nothing to do
  EXPECT(gims_sinkhorn_plan_ex(pr.data(), (int)pr.size(), 100, GIMS_OT_STREAMED) == 0);           // streamed on request, whatever the sizes
  EXPECT(gims_sinkhorn_match_ex(nullptr, 0, 1.f, 10, 0.2f, nullptr, 0, GIMS_OT_STREAMED, nullptr) == GIMS_EINVAL);
  EXPECT(gims_train_loss(nullptr, 0, nullptr, 0, 1.f, 0.45f, 1.f, nullptr, nullptr, nullptr, nullptr) == GIMS_EINVAL);
  EXPECT(gims_ingest_images(nullptr, 0, 0, 0, nullptr, 0, nullptr, nullptr, nullptr) == GIMS_EINVAL);
  EXPECT(gims_pack_graphs(nullptr, 0, 0, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr) == GIMS_EINVAL);
  EXPECT(gims_events_create(0, nullptr) == GIMS_EINVAL);
  EXPECT(gims_ops_graph_create(nullptr, 0, nullptr, nullptr) == GIMS_EINVAL);
  // ---- training-step entry points: size queries over ragged shapes, argument checks, the split-K / alignment planning of gims_gemm_f32
  {
    gims_gemm g; memset(&g, 0, sizeof(g));
    EXPECT(gims_gemm_f32(&g, nullptr) == GIMS_EINVAL);
    EXPECT(gims_gemm_f32(nullptr, nullptr) == GIMS_EINVAL);
    g.a = (const float*)0x1000; g.b = (const float*)0x2000; g.c = (float*)0x3000; g.m = 5; g.n = 7; g.k = 3; g.batch = 1; g.lda = 2; g.ldb = 3; g.ldc = 7;
    g.precision = GIMS_PREC_BF16X6;
    EXPECT(gims_gemm_f32(&g, nullptr) == GIMS_EINVAL);            // lda smaller than the row it strides
    g.lda = 3; g.precision = GIMS_PREC_F32;
    EXPECT(gims_gemm_f32(&g, nullptr) == GIMS_EINVAL);            // unsupported precision
    gims_segments sg; memset(&sg, 0, sizeof(sg));
    EXPECT(gims_batchnorm_workspace_floats(&sg, 32) == 0);
    sg.n = 8;
    for (int i = 0; i < 8; ++i) { sg.off[i] = 300 * i; sg.rows[i] = 1 + 37 * i; }
    EXPECT(gims_batchnorm_workspace_floats(&sg, 512) > 0);
    sg.n = 9;
    EXPECT(gims_batchnorm_workspace_floats(&sg, 512) == 0);
    EXPECT(gims_batchnorm_train_forward((const float*)0x1000, 512, 512, &sg, (const float*)0x1000, (const float*)0x1000, 1e-5f, 0.1f, nullptr, nullptr,
                                        (float*)0x1000, (float*)0x1000, 512, 1, (float*)0x1000, nullptr) == GIMS_EINVAL);
    EXPECT(gims_colsum_workspace_floats(4097, 768) == 64 + (size_t)33 * 768);
    EXPECT(gims_colsum((const float*)0x1001, 512, 10, 512, 0.f, (float*)0x1000, (float*)0x1000, nullptr) == GIMS_EINVAL);   // unaligned
    EXPECT(gims_softmax_rows(nullptr, 0, 1, 1, 1, 0, nullptr) == GIMS_EINVAL);
    EXPECT(gims_layernorm_backward(nullptr, 0, nullptr, 0, 1, 600, nullptr, nullptr, 1e-6f, 1, nullptr, 0, nullptr, nullptr, nullptr) == GIMS_EINVAL);
    EXPECT(gims_head_pack(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 256, 4, 0, nullptr) == GIMS_EINVAL);
    EXPECT(gims_sage_mean_transposed(nullptr, 0, nullptr, nullptr, 1, 30, nullptr, 0, nullptr) == GIMS_EINVAL);
    for (int n : {1, 63, 2048, 4097}) {
      gims_ot_problem q; memset(&q, 0, sizeof(q)); q.n = n; q.m = n + 3; q.ld = (q.m + 3) / 4 * 4;
      EXPECT(gims_sinkhorn_backward_workspace_bytes(&q, 1) > 0);
      EXPECT(gims_sinkhorn_history_floats(q.n, q.m, 100) == (size_t)101 * (size_t)(q.n + q.m + 2));
    }
    EXPECT(gims_sinkhorn_backward_workspace_bytes(nullptr, 2) == 0);
    EXPECT(gims_sinkhorn_backward(nullptr, 0, 1.f, 10, nullptr, nullptr, nullptr, nullptr, 0, nullptr) == GIMS_EINVAL);
    EXPECT(gims_sinkhorn_history(nullptr, 0, 1.f, 10, nullptr, nullptr, 0, nullptr) == GIMS_EINVAL);
  }
  {   // ---- fused Adam (round 3): table validation, and the multi-launch split of > 80 tensors
    gims_adam_group g; memset(&g, 0, sizeof(g));
    g.lr = 1e-3; g.beta1 = 0.9; g.beta2 = 0.999; g.eps = 1e-8; g.weight_decay = 0.0; g.step = 1;
    gims_adam_tensor t1; memset(&t1, 0, sizeof(t1));
    EXPECT(gims_adam_step(nullptr, 0, nullptr, 0, nullptr) == GIMS_OK);                 // nothing to do
    EXPECT(gims_adam_step(&t1, 1, &g, 9, nullptr) == GIMS_EINVAL);                      // more than 8 hyper-parameter sets
    t1.n = 16; t1.group = 0;
    EXPECT(gims_adam_step(&t1, 1, &g, 1, nullptr) == GIMS_EINVAL);                      // null tensor pointers
    t1.param = (float*)0x1000; t1.grad = (const float*)0x2000; t1.exp_avg = (float*)0x3000; t1.exp_avg_sq = (float*)0x4000; t1.group = 1;
    EXPECT(gims_adam_step(&t1, 1, &g, 1, nullptr) == GIMS_EINVAL);                      // group index out of range
    t1.group = 0; g.step = 0;
    EXPECT(gims_adam_step(&t1, 1, &g, 1, nullptr) == GIMS_EINVAL);                      // step counts from 1
    g.step = 3; g.beta2 = 1.0;
    EXPECT(gims_adam_step(&t1, 1, &g, 1, nullptr) == GIMS_EINVAL);                      // beta2 < 1
    g.beta2 = 0.999; t1.n = (int64_t)1 << 31;
    EXPECT(gims_adam_step(&t1, 1, &g, 1, nullptr) == GIMS_EINVAL);                      // element count beyond 2^31
    std::vector<gims_adam_tensor> many(300, t1);                                        // host-side chunking over 80-tensor launches; every launch
    for (size_t i = 0; i < many.size(); ++i) many[i].n = (i % 7 == 0) ? 0 : 5 + (int64_t)i;   // fails cleanly here (no device), empty tensors skipped
    const int rc_adam = gims_adam_step(many.data(), (int32_t)many.size(), &g, 1, nullptr);
    EXPECT(rc_adam == GIMS_OK || rc_adam == GIMS_EHIP);
  }
  {   // ---- the per-keypoint affine map on its own (round 3): empty input is a no-op, bad pointers are refused
    const int rc0 = gims_patch_affine(nullptr, nullptr, 0, nullptr, nullptr, nullptr);
    EXPECT(rc0 == GIMS_OK || rc0 == GIMS_EINVAL);
    EXPECT(gims_patch_affine(nullptr, nullptr, 5, nullptr, nullptr, nullptr) == GIMS_EINVAL);
  }
  // ---- a call that reaches the HIP runtime: no device in this container -> a clean GIMS_EHIP / error string, no crash
  char host_table[64] = {0};
  const int rc = gims_upload_table(host_table, sizeof(host_table), (void*)0x1000, nullptr);
  EXPECT(rc == GIMS_OK || rc == GIMS_EHIP || rc == GIMS_EINVAL);
  printf("asan host harness: %d checks passed\n", checks);
  return 0;
}

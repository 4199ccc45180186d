"""ctypes binding of libgims_hip.so (C ABI declared in include/gims_hip.h).

PyTorch is plumbing only: tensors own the device memory, ``torch.cuda.current_stream()`` provides the
hipStream_t.  There is NO CPU fallback: if the shared library is missing or a call fails this module
raises -- the product path never routes around the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgims_hip.so")

PREC_F32, PREC_BF16X3, PREC_BF16X6 = 0, 1, 2
LINEAR_UPPER = 1
LINEAR_HI_ONLY = 2
LINEAR_CONV3 = 8
LINEAR_OUT_F16 = 16      # out_bf16 receives IEEE half (saturated) instead of bf16
ACT_NONE, ACT_RELU = 0, 1


class GimsHipError(RuntimeError):
    pass


class AttnGuard(C.Structure):
    """gims_attn_guard (include/gims_hip.h): a launch that only runs when the statistic of the launch before it asks for the redo."""
    _fields_ = [("stat", C.c_void_p), ("mean_thr", C.c_double), ("tail_thr", C.c_double), ("range_limit", C.c_double),
                ("n_heads", C.c_int32), ("kind", C.c_int32), ("max_thr", C.c_double)]


GUARD_PEAKED, GUARD_RANGE = 1, 2


def attn_guard(stat, kind, n_heads, mean_thr=0.0, tail_thr=0.0, range_limit=0.0, max_thr=0.0) -> AttnGuard:
    assert stat.dtype == torch.int64 and stat.is_cuda and stat.is_contiguous() and stat.numel() >= 4 * (n_heads + 1)
    return AttnGuard(stat.data_ptr(), float(mean_thr), float(tail_thr), float(range_limit), int(n_heads), int(kind), float(max_thr))


class LinearArgs(C.Structure):
    _fields_ = [("a0", C.c_void_p), ("lda0", C.c_int64), ("a1", C.c_void_p), ("lda1", C.c_int64),
                ("w", C.c_void_p), ("w_lo", C.c_void_p), ("ldw", C.c_int64), ("bias", C.c_void_p),
                ("residual", C.c_void_p), ("out_f32", C.c_void_p), ("ldc", C.c_int64),
                ("out_bf16", C.c_void_p), ("ldc_bf16", C.c_int64), ("m", C.c_int32), ("n", C.c_int32),
                ("k", C.c_int32), ("k0", C.c_int32), ("act", C.c_int32), ("precision", C.c_int32),
                ("scale", C.c_float), ("a0_lo", C.c_void_p), ("a1_lo", C.c_void_p), ("out_hi", C.c_void_p),
                ("out_lo", C.c_void_p), ("ld_split", C.c_int64), ("flags", C.c_int32),
                ("conv_h", C.c_int32), ("conv_w", C.c_int32), ("conv_stride", C.c_int32), ("conv_reserved", C.c_int32),
                ("guard", AttnGuard), ("range_stat", C.c_void_p)]


class AttnArgs(C.Structure):
    _fields_ = [("qkv", C.c_void_p), ("ld", C.c_int64), ("q_col", C.c_int32), ("k_col", C.c_int32), ("v_col", C.c_int32),
                ("problems", C.c_void_p), ("n_problems", C.c_int32), ("max_n_q", C.c_int32), ("n_heads", C.c_int32),
                ("out", C.c_void_p), ("ld_out", C.c_int64), ("out_hi", C.c_void_p), ("out_lo", C.c_void_p),
                ("ld_split", C.c_int64), ("flags", C.c_int32), ("stat", C.c_void_p), ("guard", AttnGuard)]


class AuxArgs(C.Structure):
    """gims_aux_args: one of the small encoder-stage kernels as an op of gims_run_ops (fn = AUX_*; p / i in the entry point's argument order)."""
    _fields_ = [("fn", C.c_int32), ("reserved", C.c_int32), ("p", C.c_void_p * 6), ("i", C.c_int64 * 4)]


AUX_SPLIT_SPL32, AUX_SAGE_MEAN_SPLIT, AUX_KENC_FIRST = 0, 1, 2


class _OpU(C.Union):
    _fields_ = [("lin", LinearArgs), ("att", AttnArgs), ("aux", AuxArgs)]


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("u", _OpU)]


class AgcImage(C.Structure):
    _fields_ = [("kpts", C.c_void_p), ("desc", C.c_void_p), ("ldd", C.c_int64), ("n", C.c_int32), ("d", C.c_int32),
                ("kept", C.c_void_p), ("indptr", C.c_void_p), ("indices", C.c_void_p), ("max_edges_dir", C.c_int32),
                ("info", C.c_void_p)]


class PackImage(C.Structure):
    _fields_ = [("kpts", C.c_void_p), ("desc", C.c_void_p), ("ldd", C.c_int64), ("score", C.c_void_p),
                ("kept", C.c_void_p), ("indptr", C.c_void_p), ("indices", C.c_void_p),
                ("n_kept", C.c_int32), ("n_edges", C.c_int32), ("row_off", C.c_int32), ("edge_off", C.c_int32)]


class IngestImage(C.Structure):
    _fields_ = [("kpts", C.c_void_p), ("desc", C.c_void_p), ("ldd", C.c_int64), ("score", C.c_void_p), ("n", C.c_int32),
                ("row_off", C.c_int32)]


class OtProblem(C.Structure):
    _fields_ = [("scores", C.c_void_p), ("ld", C.c_int64), ("n", C.c_int32), ("m", C.c_int32),
                ("matches0", C.c_void_p), ("matches1", C.c_void_p), ("mscores0", C.c_void_p),
                ("mscores1", C.c_void_p), ("uv", C.c_void_p)]


class PyrLevel(C.Structure):
    _fields_ = [("offset", C.c_int64), ("h", C.c_int32), ("w", C.c_int32)]


class LossPair(C.Structure):
    _fields_ = [("scores", C.c_void_p), ("ld", C.c_int64), ("n", C.c_int32), ("m", C.c_int32), ("uv", C.c_void_p),
                ("kept0", C.c_void_p), ("kept1", C.c_void_p)]


class Gemm(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64), ("ldr", C.c_int64),
                ("sa", C.c_int64), ("sb", C.c_int64), ("sc", C.c_int64), ("sr", C.c_int64),
                ("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("batch", C.c_int32),
                ("ta", C.c_int32), ("tb", C.c_int32), ("act", C.c_int32), ("flags", C.c_int32),
                ("alpha", C.c_float), ("beta", C.c_float), ("work", C.c_void_p), ("work_floats", C.c_int64), ("splits", C.c_int32), ("precision", C.c_int32)]


class TrainAttnProblem(C.Structure):
    _fields_ = [("q_off", C.c_int32), ("nq", C.c_int32), ("k_off", C.c_int32), ("nk", C.c_int32)]


class TrainAttnArgs(C.Structure):
    _fields_ = [("qkv", C.c_void_p), ("ld", C.c_int64), ("rows", C.c_int64), ("d", C.c_int32), ("heads", C.c_int32), ("scale", C.c_float),
                ("n_problems", C.c_int32), ("problems", C.POINTER(TrainAttnProblem)), ("o", C.c_void_p), ("ldo", C.c_int64), ("lse", C.c_void_p),
                ("d_o", C.c_void_p), ("lddo", C.c_int64), ("d_qkv", C.c_void_p), ("lddq", C.c_int64), ("work", C.c_void_p), ("work_floats", C.c_size_t), ("reverse_precision", C.c_int32)]


class Segments(C.Structure):
    _fields_ = [("n", C.c_int32), ("off", C.c_int32 * 8), ("rows", C.c_int32 * 8)]


class AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("n", C.c_int64),
                ("group", C.c_int32), ("reserved", C.c_int32)]


class AdamGroup(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("weight_decay", C.c_double),
                ("step", C.c_int64)]


class EvalPair(C.Structure):
    _fields_ = [("kpts0", C.c_void_p), ("kpts1", C.c_void_p), ("matches0", C.c_void_p), ("mscores0", C.c_void_p),
                ("n0", C.c_int32), ("n1", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("h_gt", C.c_float * 9),
                ("gt0", C.c_void_p), ("inlier", C.c_void_p), ("record", C.c_void_p), ("homographies", C.c_void_p)]


_SIGNATURES = {
    "gims_abi_version": (C.c_int, []),
    "gims_last_error": (C.c_char_p, []),
    "gims_stream_sync": (C.c_int, [C.c_void_p]),
    "gims_upload_table": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "gims_linear": (C.c_int, [C.POINTER(LinearArgs), C.c_void_p]),
    "gims_linear_put": (C.c_int, [C.POINTER(LinearArgs), C.c_void_p, C.c_void_p]),
    "gims_linear_put_many": (C.c_int, [C.POINTER(LinearArgs), C.c_int32, C.c_void_p, C.c_void_p]),
    "gims_linear_batch": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "gims_split_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gims_ch_frn_stats": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "gims_ch_pool_hw": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gims_ch_frn_from_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "gims_ch_gates": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 11),
    "gims_ch_apply": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 7 + [C.c_int64, C.c_void_p]),
    "gims_ch_im2col3": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "gims_ch_dwconv3": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_float,
                                  C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gims_ch_gate_pw_pw": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 8),
    "gims_ch_input_block": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gims_ch_frn_block": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gims_ch_conv_block": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gims_ch_conv_block_first": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gims_ch_sandglass": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gims_ch_l2norm": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "gims_ch_relu6": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "gims_run_ops": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "gims_run_ops_timed": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "gims_events_create": (C.c_int, [C.c_int32, C.c_void_p]),
    "gims_events_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gims_events_elapsed": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "gims_events_destroy": (C.c_int, [C.c_void_p, C.c_int32]),
    "gims_ops_graph_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "gims_ops_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gims_ops_graph_destroy": (C.c_int, [C.c_void_p]),
    "gims_split_spl3": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "gims_split_spl32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "gims_attention": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                 C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                 C.c_int32, C.c_void_p]),
    "gims_attention_ex": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gims_attention_stat": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_int32, C.c_void_p, C.c_void_p]),
    "gims_kenc_first": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                  C.c_void_p, C.c_int64, C.c_void_p]),
    "gims_kenc_first_linear": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_void_p, C.c_int64, C.c_void_p]),
    "gims_layernorm_act": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_int32,
                                     C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gims_sage_mean": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                 C.c_void_p, C.c_int64, C.c_void_p]),
    "gims_sage_mean_split": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                 C.c_void_p, C.c_int64, C.c_void_p]),
    "gims_pair_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "gims_gather_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                   C.c_int64, C.c_void_p]),
    "gims_agc_workspace_bytes": (C.c_size_t, [C.POINTER(AgcImage), C.c_int32]),
    "gims_agc_workspace_bytes_ex": (C.c_size_t, [C.POINTER(AgcImage), C.c_int32, C.c_int32]),
    "gims_agc_max_keypoints": (C.c_int32, []),
    "gims_agc_build": (C.c_int, [C.POINTER(AgcImage), C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_void_p,
                                 C.c_size_t, C.c_void_p]),
    "gims_agc_build_ex": (C.c_int, [C.POINTER(AgcImage), C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_void_p,
                                    C.c_size_t, C.c_void_p]),
    "gims_ingest_images": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "gims_pack_graphs": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                   C.c_void_p]),
    "gims_sinkhorn_workspace_bytes": (C.c_size_t, [C.POINTER(OtProblem), C.c_int32]),
    "gims_sinkhorn_plan": (C.c_int, [C.POINTER(OtProblem), C.c_int32, C.c_int32]),
    "gims_sinkhorn_match": (C.c_int, [C.POINTER(OtProblem), C.c_int32, C.c_float, C.c_int32, C.c_float,
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    "gims_sinkhorn_plan_ex": (C.c_int, [C.POINTER(OtProblem), C.c_int32, C.c_int32, C.c_int32]),
    "gims_sinkhorn_rescues": (C.c_int64, []),
    "gims_attention_launch_counts": (C.c_int, [C.POINTER(C.c_uint64), C.c_int32, C.c_int32]),
    "gims_sinkhorn_match_ex": (C.c_int, [C.POINTER(OtProblem), C.c_int32, C.c_float, C.c_int32, C.c_float,
                                         C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "gims_eval_workspace_bytes": (C.c_size_t, [C.POINTER(EvalPair), C.c_int32, C.c_int32]),
    "gims_eval_pairs": (C.c_int, [C.POINTER(EvalPair), C.c_int32, C.c_float, C.c_int32, C.c_float, C.c_int32, C.c_uint64,
                                  C.c_void_p, C.c_size_t, C.c_void_p]),
    "gims_pyramid_layout": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_size_t)]),
    "gims_pyramid_build": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gims_patch_extract": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gims_patch_affine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gims_sinkhorn_history_floats": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "gims_sinkhorn_history": (C.c_int, [C.POINTER(OtProblem), C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gims_sinkhorn_backward_workspace_bytes": (C.c_size_t, [C.POINTER(OtProblem), C.c_int32]),
    "gims_sinkhorn_backward": (C.c_int, [C.POINTER(OtProblem), C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_void_p]),
    "gims_train_loss_grad": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "gims_train_loss": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "gims_gemm_f32": (C.c_int, [C.POINTER(Gemm), C.c_void_p]),
    "gims_batchnorm_workspace_floats": (C.c_size_t, [C.POINTER(Segments), C.c_int32]),
    "gims_batchnorm_train_forward": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(Segments), C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "gims_batchnorm_train_backward": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(Segments), C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gims_layernorm_backward": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_int32,
                                          C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gims_softmax_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]),
    "gims_softmax_rows_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]),
    "gims_train_attention_workspace_floats": (C.c_size_t, [C.c_int64, C.c_int32]),
    "gims_train_attention_forward": (C.c_int, [C.POINTER(TrainAttnArgs), C.c_void_p]),
    "gims_train_attention_backward": (C.c_int, [C.POINTER(TrainAttnArgs), C.c_void_p]),
    "gims_colsum_workspace_floats": (C.c_size_t, [C.c_int64, C.c_int32]),
    "gims_colsum": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gims_elementwise": (C.c_int, [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_float,
                                   C.c_void_p]),
    "gims_permute3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_int64] * 6 + [C.c_int32, C.c_void_p]),
    "gims_head_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "gims_sage_mean_transposed": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "gims_normalize_keypoints": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "gims_adam_step": (C.c_int, [C.POINTER(AdamTensor), C.c_int32, C.POINTER(AdamGroup), C.c_int32, C.c_void_p]),
    "gims_ot_matrix": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def load(path: str | None = None):
    """dlopen the kernel library and type its entry points.  Raises if it is not there."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise GimsHipError(f"{p} is missing: build it with `python -m gims_amd.build` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(p)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.gims_abi_version() != ABI_VERSION:
        raise GimsHipError(f"ABI version mismatch: library {lib.gims_abi_version()} != binding {ABI_VERSION} (rebuild: python -m gims_amd.build)")
    if path is None:
        _lib = lib
    return lib


GIMS_OK, GIMS_EINVAL, GIMS_EHIP, GIMS_ENUMERIC = 0, -1, -2, -3          # include/gims_hip.h
ABI_VERSION = 2                     # GIMS_ABI_VERSION of the header these ctypes mirrors were written against


def _check(rc: int, what: str):
    if rc != 0:
        msg = load().gims_last_error()
        raise GimsHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


import threading  # noqa: E402

_TLS = threading.local()    # per thread: .pin = the raw hipStream_t pinned by pinned_stream(), .gemm_prec = gemm()'s default precision


def _stream() -> int:
    pin = getattr(_TLS, "pin", None)
    return pin if pin is not None else torch.cuda.current_stream().cuda_stream


class on_stream:
    """``with on_stream(handle):`` -- launches of the calling thread go to the raw HIP stream `handle` inside the block (the side stream of
    the training step's parameter-gradient work).  Ordering against the surrounding stream and the lifetime of the tensors involved are
    the caller's business (events; references held until the streams have joined)."""

    def __init__(self, handle: int):
        self._h = int(handle)

    def __enter__(self):
        self._prev = getattr(_TLS, "pin", None)
        _TLS.pin = self._h
        return self

    def __exit__(self, *exc):
        _TLS.pin = self._prev
        return False


class pinned_stream:
    """``with pinned_stream():`` -- look the current torch stream up ONCE for a run of launches (torch.cuda.current_stream() costs a
    few microseconds per call; a training step makes ~1 800 launches).  The pin belongs to the calling THREAD (other threads
    keep resolving their own current stream), and the stream must not be switched inside the block: leaving it with a
    different current stream raises instead of having launched on the wrong one silently."""

    def __enter__(self):
        self._prev = getattr(_TLS, "pin", None)
        _TLS.pin = torch.cuda.current_stream().cuda_stream
        return self

    def __exit__(self, exc_type, *exc):
        pin, _TLS.pin = _TLS.pin, self._prev
        if exc_type is None and torch.cuda.current_stream().cuda_stream != pin:
            raise GimsHipError("the current CUDA stream changed inside a pinned_stream() block: launches went to the stream that was current on entry")
        return False


def _p(t) -> int | None:
    return None if t is None else t.data_ptr()


_NP2TORCH = {"float32": torch.float32, "int32": torch.int32, "int64": torch.int64, "uint8": torch.uint8}


def upload(arr, device="cuda", out=None) -> torch.Tensor:
    """Small host table (numpy array) -> device tensor, asynchronously and in stream order (gims_upload_table): the
    bytes ride in kernel arguments, so unlike ``torch.tensor(..., device=...)`` / ``.to(device)`` from pageable memory
    the calling thread never waits for the stream to drain."""
    import numpy as np
    a = np.ascontiguousarray(arr)
    dt = _NP2TORCH[a.dtype.name]
    nbytes = a.nbytes
    if out is not None:      # caller-owned uint8 arena (stable address from call to call); stream order protects earlier readers
        assert out.dtype == torch.uint8 and out.numel() >= nbytes + 16
        buf = out
    else:
        buf = torch.empty(((nbytes + 15) // 16 * 16 + 16,), dtype=torch.uint8, device=device)   # caching allocator: 512-B aligned
    if nbytes:
        _check(load().gims_upload_table(a.ctypes.data, nbytes, buf.data_ptr(), _stream()), "gims_upload_table")
    return buf[:nbytes].view(dt).view(a.shape)


def _dev(t: torch.Tensor, dtype=None):
    if not t.is_cuda:
        raise GimsHipError("expected a device tensor")
    if dtype is not None and t.dtype != dtype:
        raise GimsHipError(f"expected {dtype}, got {t.dtype}")
    return t


def linear_args(a0, w, *, bias=None, a1=None, w_lo=None, residual=None, out=None, out_bf16=None, act=ACT_NONE,
                precision=PREC_F32, scale=1.0, n=None, spl=False, out_split=None, flags=0, conv=None, m=None, guard=None, range_stat=None):
    """Build the C struct.
    spl=False: a0/a1 f32 [m,k*]; w f32 [n,K] (PREC_F32) or bf16 hi plane with w_lo (PREC_BF16X3).
    spl=True : a0/a1/w are SPL32 bf16 buffers [rows, 2*k] (see include/gims_hip.h), precision BF16X3.
    out_split: SPL32 bf16 buffer [m, >= 2n] receiving the result split into hi/lo."""
    m = a0.shape[0] if m is None else m
    if conv is not None:
        # 3x3 convolution read straight from an SPL32 NHWC activation: a0 = pixel rows [n*h*w, 2C], a1 = 128 zero bytes,
        # w = SPL32 [n_out, 2*9C], conv = (h, w, stride), m = output pixels
        assert spl and precision == PREC_BF16X3 and a1 is not None and m is not None
        k0 = k = w.shape[1] // 2
        assert k % 9 == 0 and a0.shape[1] == 2 * (k // 9)
        a0_lo, a1_lo, w_lo = a0[:, 32:], None, w[:, 32:]
        args = LinearArgs(_p(a0), a0.stride(0), _p(a1), 0, _p(w), _p(w_lo), w.stride(0), _p(bias), None, _p(out),
                          out.stride(0) if out is not None else 0, None, 0, m, w.shape[0] if n is None else n, k, k0, act, precision, float(scale),
                          _p(a0_lo), None, None, None, 0, int(flags) | LINEAR_CONV3, int(conv[0]), int(conv[1]), int(conv[2]), 0)
        return args
    if precision == PREC_BF16X6:
        # a0 / w are SPL3 bf16 buffers [rows, 3k] (split_spl3); plain f32 output only
        assert a0.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and a1 is None and not spl
        k0 = k = a0.shape[1] // 3
        assert w.shape[1] == 3 * k, (w.shape, k)
        a0_lo = a1_lo = w_lo = None
    elif spl:
        assert a0.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and precision == PREC_BF16X3
        k0 = a0.shape[1] // 2
        k = k0 + (a1.shape[1] // 2 if a1 is not None else 0)
        assert w.shape[1] == 2 * k, (w.shape, k)
        # (the lo planes sit 32 elements = 64 bytes behind the hi planes of the same buffers: plain pointer arithmetic -- a sliced view per
        # operand cost a single-pair forward() ~8 us of Python per GEMM launch)
        p_a0, p_a1, p_w = a0.data_ptr(), (a1.data_ptr() if a1 is not None else None), w.data_ptr()
        return _linear_struct(p_a0, a0.stride(0), p_a1, a1.stride(0) if a1 is not None else 0, p_w, p_w + 64, w, bias, residual, out, out_bf16, m, n,
                              k, k0, act, precision, scale, p_a0 + 64, (p_a1 + 64) if p_a1 is not None else None, out_split, flags, guard, range_stat)
    else:
        _dev(a0, torch.float32)
        k0 = a0.shape[1]
        k = k0 + (a1.shape[1] if a1 is not None else 0)
        assert w.shape[1] == k, (w.shape, k)
        a0_lo = a1_lo = None
    return _linear_struct(_p(a0), a0.stride(0), _p(a1), a1.stride(0) if a1 is not None else 0, _p(w), _p(w_lo), w, bias, residual, out, out_bf16, m, n,
                          k, k0, act, precision, scale, _p(a0_lo), _p(a1_lo), out_split, flags, guard, range_stat)


def _linear_struct(p_a0, lda0, p_a1, lda1, p_w, p_w_lo, w, bias, residual, out, out_bf16, m, n, k, k0, act, precision, scale, p_a0_lo, p_a1_lo, out_split,
                   flags, guard, range_stat) -> LinearArgs:
    n = w.shape[0] if n is None else n
    assert w.stride(1) == 1
    if residual is not None:
        assert out is not None and residual.stride(0) == out.stride(0)
    if out_split is not None:
        assert out_split.dtype == torch.bfloat16 and out_split.shape[1] >= 2 * n and out_split.stride(1) == 1
    return LinearArgs(p_a0, lda0, p_a1, lda1, p_w, p_w_lo, w.stride(0), _p(bias), _p(residual), _p(out),
                      out.stride(0) if out is not None else 0, _p(out_bf16),
                      out_bf16.stride(0) if out_bf16 is not None else 0, m, n, k, k0, act, precision, float(scale),
                      p_a0_lo, p_a1_lo, _p(out_split), (out_split.data_ptr() + 64) if out_split is not None else None,
                      out_split.stride(0) if out_split is not None else 0, int(flags), 0, 0, 0, 0,
                      guard if guard is not None else AttnGuard(), _p(range_stat))


def linear_batch(arg_list, dev_args: torch.Tensor, precision=PREC_F32):
    """Many independent problems in one launch; dev_args: uint8 device scratch of >= len * sizeof(LinearArgs)."""
    lib = load()
    sz = C.sizeof(LinearArgs)
    assert dev_args.numel() * dev_args.element_size() >= sz * len(arg_list)
    st = _stream()
    arr = (LinearArgs * len(arg_list))(*arg_list)
    _check(lib.gims_linear_put_many(arr, len(arg_list), dev_args.data_ptr(), st), "gims_linear_put_many")
    _check(lib.gims_linear_batch(dev_args.data_ptr(), len(arg_list), max(a.m for a in arg_list), max(a.n for a in arg_list),
                                 precision, st), "gims_linear_batch")


def linear(a0, w, *, out=None, out_bf16=None, out_split=None, **kw):
    """C = act(scale * [a0 | a1] @ w[:n].T + bias) (+ residual); see linear_args / include/gims_hip.h."""
    lib = load()
    if out is None and out_bf16 is None and out_split is None:
        n = kw.get("n") or w.shape[0]
        out = torch.empty((a0.shape[0], n), dtype=torch.float32, device=a0.device)
    args = linear_args(a0, w, out=out, out_bf16=out_bf16, out_split=out_split, **kw)
    _check(lib.gims_linear(C.byref(args), _stream()), "gims_linear")
    return out if out is not None else (out_bf16 if out_bf16 is not None else out_split)


def split_spl32(x: torch.Tensor, out: torch.Tensor | None = None):
    """f32 [rows, k] -> SPL32 bf16 [rows, 2k] (32 hi | 32 lo per 32-channel block)."""
    lib = load()
    rows, k = x.shape
    assert x.stride(1) == 1 and k % 32 == 0
    if out is None:
        out = torch.empty((rows, 2 * k), dtype=torch.bfloat16, device=x.device)
    _check(lib.gims_split_spl32(_p(_dev(x, torch.float32)), x.stride(0), _p(out), out.stride(0), rows, k, _stream()), "gims_split_spl32")
    return out


def op_linear(args: LinearArgs) -> Op:
    o = Op()
    o.kind = 0
    o.u.lin = args
    return o


def _attn_flags(q_prescaled, x3, f16, no_range=False):
    assert not (x3 and f16)
    return (1 if q_prescaled else 0) | (ATTN_X3 if x3 else 0) | (ATTN_F16 if f16 else 0) | (ATTN_NO_RANGE if no_range else 0)


def op_attention(qkv, problems, max_n_q, n_heads, out=None, q_col=0, k_col=256, v_col=512, out_split=None, q_prescaled=False, x3=False,
                 stat=None, f16=False, guard=None, no_range=False) -> Op:
    o = Op()
    o.kind = 1
    o.u.att = AttnArgs(_p(qkv), qkv.stride(0), q_col, k_col, v_col, _p(problems), problems.shape[0], max_n_q, n_heads, _p(out),
                       out.stride(0) if out is not None else 0, _p(out_split), (out_split.data_ptr() + 64) if out_split is not None else None,
                       out_split.stride(0) if out_split is not None else 0, _attn_flags(q_prescaled, x3, f16, no_range), _p(stat),
                       guard if guard is not None else AttnGuard())
    return o


def op_aux(fn: int, ptrs, ints) -> Op:
    """One small encoder-stage kernel as an op (include/gims_hip.h GIMS_OP_AUX): ptrs = tensors or raw addresses (None -> NULL), ints = integers."""
    o = Op()
    o.kind = 2
    o.u.aux.fn = int(fn)
    for k, t in enumerate(ptrs):
        o.u.aux.p[k] = t if (t is None or isinstance(t, int)) else t.data_ptr()
    for k, v in enumerate(ints):
        o.u.aux.i[k] = int(v)
    return o


def make_ops(ops):
    """A replayable launch sequence (see gims_run_ops): returns the ctypes array; keep the tensors it points to alive."""
    return (Op * len(ops))(*ops)


def run_ops(op_array, start: int = 0, count: int | None = None):
    """Replay ops[start : start + count] (default: all of them) in one call."""
    n = len(op_array) - start if count is None else count
    assert 0 <= start and start + n <= len(op_array)
    _check(load().gims_run_ops(C.byref(op_array, start * C.sizeof(Op)) if start else op_array, n, _stream()), "gims_run_ops")


class EventPool:
    """n HIP events owned by the library (gims_events_*): per-op timing of a replayed launch sequence."""

    def __init__(self, n):
        self.n = n
        self._ev = (C.c_void_p * n)()
        _check(load().gims_events_create(n, self._ev), "gims_events_create")

    def elapsed_ms(self):
        """The n - 1 intervals between consecutive events (ms); synchronise the stream first."""
        out = (C.c_float * (self.n - 1))()
        _check(load().gims_events_elapsed(self._ev, self.n, out), "gims_events_elapsed")
        return list(out)

    def __del__(self):
        try:
            load().gims_events_destroy(self._ev, self.n)
        except Exception:
            pass


def run_ops_timed(op_array, pool: EventPool):
    assert pool.n == len(op_array) + 1
    _check(load().gims_run_ops_timed(op_array, len(op_array), _stream(), pool._ev), "gims_run_ops_timed")


class OpsGraph:
    """The launch sequence as an instantiated HIP graph (gims_ops_graph_*); capture needs a non-default stream."""

    def __init__(self, op_array):
        self._ops = op_array          # keeps the host table alive
        h = C.c_void_p()
        _check(load().gims_ops_graph_create(op_array, len(op_array), _stream(), C.byref(h)), "gims_ops_graph_create")
        self._h = h

    def launch(self):
        _check(load().gims_ops_graph_launch(self._h, _stream()), "gims_ops_graph_launch")

    def __del__(self):
        try:
            if self._h:
                load().gims_ops_graph_destroy(self._h)
        except Exception:
            pass


def split_spl3(x: torch.Tensor, out: torch.Tensor | None = None):
    """f32 [rows, k] -> SPL3 bf16 [rows, 3k]: the exact three-way split x = a1 + a2 + a3 (32 a1 | 32 a2 | 32 a3 per
    32-channel block), the operand layout of PREC_BF16X6."""
    lib = load()
    rows, k = x.shape
    assert x.stride(1) == 1 and k % 32 == 0
    if out is None:
        out = torch.empty((rows, 3 * k), dtype=torch.bfloat16, device=x.device)
    _check(lib.gims_split_spl3(_p(_dev(x, torch.float32)), x.stride(0), _p(out), out.stride(0), rows, k, _stream()), "gims_split_spl3")
    return out


def spl32_planes(buf: torch.Tensor):
    """Decode an SPL32 buffer [rows, 2k] into (hi, lo) bf16 tensors [rows, k] (test / debug helper)."""
    rows, k2 = buf.shape
    v = buf.reshape(rows, k2 // 64, 2, 32)
    return v[:, :, 0, :].reshape(rows, k2 // 2), v[:, :, 1, :].reshape(rows, k2 // 2)


def split_bf16(x: torch.Tensor):
    lib = load()
    x = x.contiguous()
    hi = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    lo = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _check(lib.gims_split_bf16(_p(_dev(x, torch.float32)), _p(hi), _p(lo), x.numel(), _stream()), "gims_split_bf16")
    return hi, lo


ATTN_Q_SCALE = 0.125 * 1.4426950408889634      # log2(e) / sqrt(64): what q_prescaled=True expects folded into Q


ATTN_X3 = 2
ATTN_NO_RANGE = 8   # a measured launch leaves the range row alone (the projection reported it: linear_args(range_stat=...))
ATTN_F16 = 4        # qkv holds IEEE half (gims_linear with LINEAR_OUT_F16); v_mfma_f32_32x32x16_f16 kernels


ATTN_STAT_SCALE = float(1 << 24)       # fixed point of the row maxima in gims_attention_stat's accumulator


def attention(qkv: torch.Tensor, problems: torch.Tensor, max_n_q: int, n_heads: int, out=None,
              q_col=0, k_col=256, v_col=512, out_split=None, q_prescaled=False, x3=False, stat=None, f16=False, guard=None, no_range=False):
    """qkv bf16 [rows, ld]; problems int32 [P,4] (q_off, n_q, kv_off, n_kv) on device; out f32 [rows, ld_out]
    and/or out_split = SPL32 bf16 buffer [rows, >= 512].  q_prescaled: Q already carries ATTN_Q_SCALE.
    x3: qkv is the SPL32 split-bf16 buffer [rows, >= 1536] of the 3-pass projection (GIMS_ATTN_X3).
    f16: the 16-bit values of qkv are IEEE half, not bf16 (GIMS_ATTN_F16; the tensor's dtype stays torch.bfloat16: raw storage).
    stat: int64 [n_heads + 1, 4] accumulator of the softmax peakedness and the operand range (gims_attention_stat; zero it before the
    first use)."""
    lib = load()
    assert qkv.dtype == torch.bfloat16 and problems.dtype == torch.int32 and problems.is_cuda
    if stat is not None:
        assert stat.dtype == torch.int64 and stat.is_contiguous() and stat.numel() >= 4 * (n_heads + 1) and stat.is_cuda
    if guard is not None:       # guarded launch (x3 only): a no-op unless the guard's statistic asks for the redo
        op = op_attention(qkv, problems, max_n_q, n_heads, out, q_col, k_col, v_col, out_split, q_prescaled, x3, stat, f16, guard, no_range)
        _check(lib.gims_attention_ex(C.byref(op.u.att), _stream()), "gims_attention_ex")
        return out if out is not None else out_split
    _check(lib.gims_attention_stat(_p(qkv), qkv.stride(0), q_col, k_col, v_col, _p(problems), problems.shape[0],
                                   max_n_q, n_heads, _p(out), out.stride(0) if out is not None else 0, _p(out_split),
                                   (out_split.data_ptr() + 64) if out_split is not None else None,
                                   out_split.stride(0) if out_split is not None else 0,
                                   _attn_flags(q_prescaled, x3, f16, no_range), _p(stat), _stream()),
           "gims_attention")
    return out if out is not None else out_split


def kenc_first(kpts, norm3, seg_of_row, w1, b1, out, relu=True):
    lib = load()
    c1 = w1.shape[0]
    fn = lib.gims_kenc_first if relu else lib.gims_kenc_first_linear
    _check(fn(_p(_dev(kpts, torch.float32)), _p(norm3), _p(seg_of_row), _p(w1), _p(b1), c1,
              _p(out), kpts.shape[0], _stream()), "gims_kenc_first")
    return out


def layernorm_act(x, a2, b2, out=None, out_split=None, act=ACT_RELU, eps=1e-6):
    """LayerNorm over the channels of every row (unbiased std, eps on the std; gmatcher.py:74-85) + activation.
    out: f32 [rows, c] (may be x itself) and/or out_split: SPL32 bf16 [rows, >= 2c]."""
    lib = load()
    rows, c = x.shape
    assert x.dtype == torch.float32 and x.stride(1) == 1 and (out is not None or out_split is not None)
    _check(lib.gims_layernorm_act(_p(x), x.stride(0), rows, c, _p(a2), _p(b2), float(eps), act, _p(out),
                                  out.stride(0) if out is not None else 0, _p(out_split),
                                  (out_split.data_ptr() + 64) if out_split is not None else None,
                                  out_split.stride(0) if out_split is not None else 0, _stream()), "gims_layernorm_act")
    return out if out is not None else out_split


def sage_mean_split(h, indptr, indices, out_spl, n=None, c=None):
    """mean over CSR neighbours, written as SPL32 split-bf16 planes (out_spl: bf16 [rows, 2 * c])."""
    lib = load()
    n = h.shape[0] if n is None else n
    c = h.shape[1] if c is None else c
    _check(lib.gims_sage_mean_split(_p(_dev(h, torch.float32)), h.stride(0), _p(indptr), _p(indices), n, c, _p(out_spl),
                                    out_spl.stride(0), _stream()), "gims_sage_mean_split")
    return out_spl


def sage_mean(h, indptr, indices, out, n=None, c=None):
    lib = load()
    n = h.shape[0] if n is None else n
    c = h.shape[1] if c is None else c
    _check(lib.gims_sage_mean(_p(_dev(h, torch.float32)), h.stride(0), _p(indptr), _p(indices), n, c, _p(out),
                              out.stride(0), _stream()), "gims_sage_mean")
    return out


def pair_stats(matches0, scores0, table):
    """[n_pairs, 5] f32 records {pair_id, n0, n1, n_matches, mean_score} from batch-concatenated outputs; table: int32
    [n_pairs, 4] = {pair_id, n0, n1, row offset} on the device (gims_pair_stats)."""
    lib = load()
    n = table.shape[0]
    out = torch.empty((n, 5), dtype=torch.float32, device=matches0.device)
    _check(lib.gims_pair_stats(_p(_dev(matches0, torch.int64)), _p(_dev(scores0, torch.float32)), _p(_dev(table, torch.int32)), n, _p(out),
                               _stream()), "gims_pair_stats")
    return out


def gather_rows(src, idx, out):
    lib = load()
    _check(lib.gims_gather_rows(_p(_dev(src, torch.float32)), src.stride(0), _p(idx), idx.shape[0], src.shape[1],
                                _p(out), out.stride(0), _stream()), "gims_gather_rows")
    return out


def make_agc_images(items):
    """items: dicts with kpts [n,2], desc [n,d] (point-major f32), kept [n], indptr [n+1], indices [cap], info [8]."""
    arr = (AgcImage * len(items))()
    for i, it in enumerate(items):
        de = it["desc"]
        assert de.stride(1) == 1 and it["kpts"].is_contiguous()
        arr[i] = AgcImage(_p(_dev(it["kpts"], torch.float32)), _p(_dev(de, torch.float32)), de.stride(0), de.shape[0], de.shape[1],
                          _p(it["kept"]), _p(it["indptr"]), _p(it["indices"]), it["indices"].numel(), _p(it["info"]))
    return arr


def agc_max_keypoints() -> int:
    """Largest image the graph build takes (32768; include/gims_hip.h says what bounds it)."""
    return int(load().gims_agc_max_keypoints())


def agc_workspace_bytes(images, flags=None) -> int:
    """Scratch bytes of agc_build(images, ..., flags=flags); flags=None: enough for either flow (the robust one stores the half N x N matrix)."""
    nmax = max(int(im.n) for im in images)
    if nmax > agc_max_keypoints():
        raise GimsHipError(f"adaptive graph: an image has {nmax} keypoints, more than the library's limit of {agc_max_keypoints()} per image "
                           "(gims_agc_max_keypoints; the reference has none) -- reduce max_keypoints or split the image")
    if flags is None:
        return int(load().gims_agc_workspace_bytes(images, len(images)))
    return int(load().gims_agc_workspace_bytes_ex(images, len(images), int(flags)))


AGC_ROBUST = 1               # gims_agc_build_ex flags (include/gims_hip.h)
AGC_INFO_OVERFLOW, AGC_INFO_WINDOW_MISSED = 1, 2     # bits of info[7]


def agc_build(images, radius, percentile, min_size, work: torch.Tensor, flags=0):
    """Asynchronous adaptive-graph build for a batch of images; see include/gims_hip.h.  An image whose info[7] has AGC_INFO_WINDOW_MISSED
    set after the call must be rebuilt with flags=AGC_ROBUST."""
    lib = load()
    _check(lib.gims_agc_build_ex(images, len(images), float(radius), float(percentile), int(min_size), int(flags), _p(work),
                                 work.numel() * work.element_size(), _stream()), "gims_agc_build_ex")


def _upload_structs(arr, device):
    """ctypes descriptor table -> device through gims_upload_table (stream-ordered, never blocks the host)."""
    nbytes = C.sizeof(arr)
    buf = torch.empty(((nbytes + 15) // 16 * 16 + 16,), dtype=torch.uint8, device=device)
    _check(load().gims_upload_table(C.addressof(arr), nbytes, buf.data_ptr(), _stream()), "gims_upload_table")
    return buf


def ingest_images(items, d, desc_out, kpts_out, score_out):
    """items: list of IngestImage (host).  One launch for the batch."""
    lib = load()
    arr = (IngestImage * len(items))(*items)
    dev_arr = _upload_structs(arr, desc_out.device)
    _check(lib.gims_ingest_images(_p(dev_arr), len(items), max(i.n for i in items), d, _p(desc_out), desc_out.stride(0),
                                  _p(kpts_out), _p(score_out), _stream()), "gims_ingest_images")
    return dev_arr


PACK_DTYPE = None


def pack_table(images_ptrs):
    """numpy structured array mirroring gims_pack_image (one record per image) with the pointer columns filled;
    the count/offset columns are filled after the graph build's host sync (vectorised: no per-image Python work then)."""
    import numpy as np
    global PACK_DTYPE
    if PACK_DTYPE is None:
        PACK_DTYPE = np.dtype([("kpts", "<u8"), ("desc", "<u8"), ("ldd", "<i8"), ("score", "<u8"), ("kept", "<u8"),
                               ("indptr", "<u8"), ("indices", "<u8"), ("n_kept", "<i4"), ("n_edges", "<i4"),
                               ("row_off", "<i4"), ("edge_off", "<i4")])
        assert PACK_DTYPE.itemsize == C.sizeof(PackImage)
    t = np.zeros(len(images_ptrs), dtype=PACK_DTYPE)
    for i, r in enumerate(images_ptrs):
        t[i] = r + (0, 0, 0, 0)
    return t


def pack_graphs_table(table, d, feat, kpts_out, score_out, seg, indptr_out, indices_out, n_rows, n_edges):
    """pack_graphs with the descriptor table given as the numpy array of pack_table()."""
    lib = load()
    nbytes = table.nbytes
    dev_arr = torch.empty(((nbytes + 15) // 16 * 16 + 16,), dtype=torch.uint8, device=feat.device)
    st = _stream()
    _check(lib.gims_upload_table(table.ctypes.data, nbytes, dev_arr.data_ptr(), st), "gims_upload_table")
    _check(lib.gims_pack_graphs(_p(dev_arr), len(table), int(table["n_kept"].max()), max(int(table["n_edges"].max()), 1), d,
                                _p(feat), feat.stride(0), _p(kpts_out), _p(score_out), _p(seg), _p(indptr_out),
                                _p(indices_out), n_rows, n_edges, st), "gims_pack_graphs")
    return dev_arr


def pack_graphs(pack_items, d, feat, kpts_out, score_out, seg, indptr_out, indices_out, n_rows, n_edges):
    """pack_items: list of PackImage (host); uploaded as kernel arguments, then one launch for the batch."""
    lib = load()
    arr = (PackImage * len(pack_items))(*pack_items)
    dev_arr = _upload_structs(arr, feat.device)
    _check(lib.gims_pack_graphs(_p(dev_arr), len(pack_items), max(p.n_kept for p in pack_items),
                                max(max(p.n_edges for p in pack_items), 1), d, _p(feat), feat.stride(0), _p(kpts_out),
                                _p(score_out), _p(seg), _p(indptr_out), _p(indices_out), n_rows, n_edges, _stream()),
           "gims_pack_graphs")
    return dev_arr


def make_ot_problems(items):
    """items: list of dicts with scores (f32 [n, ld>=m] view), n, m, matches0/1, mscores0/1, uv tensors."""
    arr = (OtProblem * len(items))()
    for i, it in enumerate(items):
        s = it["scores"]
        arr[i] = OtProblem(_p(s), s.stride(0), it["n"], it["m"], _p(it["matches0"]), _p(it["matches1"]),
                           _p(it["mscores0"]), _p(it["mscores1"]), _p(it["uv"]))
    return arr


def sinkhorn_workspace_bytes(problems) -> int:
    return int(load().gims_sinkhorn_workspace_bytes(problems, len(problems)))


OT_STREAMED = 1      # flag of sinkhorn_plan / sinkhorn_match: never an on-chip kernel (concurrent streams on one GPU)


def sinkhorn_plan(problems, iters: int, flags: int = 0) -> int:
    """0: streamed kernels (one launch per iteration); k > 0: on-chip resident kernel in k launches."""
    return int(load().gims_sinkhorn_plan_ex(problems, len(problems), int(iters), int(flags)))


ATTN_KERNEL_KINDS = ("wave4", "split", "wave8", "wave8_f16", "x3", "x3_guarded")


def attention_launch_counts(reset: bool = False) -> dict:
    """Attention launches of this process per kernel family since the last reset (include/gims_hip.h GIMS_ATTN_KERNEL_*)."""
    buf = (C.c_uint64 * len(ATTN_KERNEL_KINDS))()
    _check(load().gims_attention_launch_counts(buf, len(ATTN_KERNEL_KINDS), int(reset)), "gims_attention_launch_counts")
    return dict(zip(ATTN_KERNEL_KINDS, (int(x) for x in buf)))


def sinkhorn_rescues() -> int:
    """On-chip solves of this process that gave up and were re-solved by a rescue path on the current device (synchronises)."""
    n = int(load().gims_sinkhorn_rescues())
    if n < 0:
        _check(GIMS_EHIP, "gims_sinkhorn_rescues")
    return n


def sinkhorn_match(problems, alpha: float, iters: int, match_threshold: float, work: torch.Tensor, flags: int = 0):
    lib = load()
    _check(lib.gims_sinkhorn_match_ex(problems, len(problems), float(alpha), int(iters), float(match_threshold), _p(work),
                                      work.numel() * work.element_size(), int(flags), _stream()), "gims_sinkhorn_match")


def ot_matrix(scores, n, m, alpha, uv):
    lib = load()
    out = torch.empty((n + 1, m + 1), dtype=torch.float32, device=scores.device)
    _check(lib.gims_ot_matrix(_p(scores), scores.stride(0), n, m, float(alpha), _p(uv), _p(out), _stream()),
           "gims_ot_matrix")
    return out


# ------------------------------------------------------------------------------------------------ evaluation
EVAL_FIELDS = ("n_valid", "n_gt", "n_correct", "n_fn", "precision", "recall", "n_inliers", "err_dlt", "err_ransac", "dlt_ok",
               "ransac_ok")


def eval_pairs(items, dist_thresh=3.0, n_iters=3, ransac_thresh=3.0, ransac_iters=3000, seed=0, work=None):
    """items: list of dicts with device tensors kpts0 [n0,2] f32, kpts1 [n1,2] f32, matches0 [n0] int64, mscores0 [n0] f32,
    h_gt (3x3 array-like), height, width, and outputs gt0 [n0] int32, inlier [n0] uint8, record [16] f32,
    homographies [18] f32.  One batched asynchronous call; see include/gims_hip.h."""
    import numpy as np
    lib = load()
    arr = (EvalPair * len(items))()
    for i, it in enumerate(items):
        k0, k1 = it["kpts0"], it["kpts1"]
        assert k0.dtype == torch.float32 and k1.dtype == torch.float32 and k0.is_contiguous() and k1.is_contiguous()
        assert it["matches0"].dtype == torch.int64 and it["mscores0"].dtype == torch.float32
        assert it["gt0"].dtype == torch.int32 and it["inlier"].dtype == torch.uint8
        h = np.asarray(it["h_gt"], dtype=np.float32).reshape(9)
        arr[i] = EvalPair(_p(k0), _p(k1), _p(it["matches0"]), _p(it["mscores0"]), k0.shape[0], k1.shape[0], int(it["height"]),
                          int(it["width"]), (C.c_float * 9)(*h.tolist()), _p(it["gt0"]), _p(it["inlier"]), _p(it["record"]),
                          _p(it["homographies"]))
    need = int(lib.gims_eval_workspace_bytes(arr, len(items), int(ransac_iters)))
    if work is None or work.numel() * work.element_size() < need:
        work = torch.empty(need, dtype=torch.uint8, device=items[0]["kpts0"].device)
    _check(lib.gims_eval_pairs(arr, len(items), float(dist_thresh), int(n_iters), float(ransac_thresh), int(ransac_iters),
                               int(seed) & 0xFFFFFFFFFFFFFFFF, _p(work), work.numel() * work.element_size(), _stream()),
           "gims_eval_pairs")
    return work


# ------------------------------------------------------------------------------------------------ CAR-HyNet ops (NHWC f32)
def ch_frn_stats(x, weight, eps, scale):
    n, h, w, c = x.shape
    _check(load().gims_ch_frn_stats(_p(_dev(x, torch.float32)), n, h * w, c, _p(weight), float(eps), _p(scale), _stream()), "gims_ch_frn_stats")
    return scale


def ch_pool_hw(x, s, b, ph, pw, rowsq=None):
    n, h, w, c = x.shape
    _check(load().gims_ch_pool_hw(_p(_dev(x, torch.float32)), n, h, w, c, _p(s), _p(b), _p(ph), _p(pw), _p(rowsq), _stream()), "gims_ch_pool_hw")


def ch_frn_from_rows(rowsq, w, weight, eps, scale):
    n, h, c = rowsq.shape
    _check(load().gims_ch_frn_from_rows(_p(rowsq), n, h, w, c, _p(weight), float(eps), _p(scale), _stream()), "gims_ch_frn_from_rows")
    return scale


def ch_gates(ph, pw, g, ah, aw, frn_scale=None, frn_bias=None):
    n, h, c = ph.shape
    w = pw.shape[1]
    _check(load().gims_ch_gates(_p(ph), _p(pw), n, h, w, c, _p(g["w1"]), _p(g["b1"]), _p(g["wh"]), _p(g["bh"]), _p(g["ww"]), _p(g["bw"]),
                                _p(frn_scale), _p(frn_bias), _p(ah), _p(aw), _stream()), "gims_ch_gates")


def ch_apply(x, s, b, ah, aw, tau, y=None, y_split=None):
    """y: f32 NHWC and / or y_split: SPL32 bf16 [n*h*w, >= 2c] pixel rows (the operand layout of the next convolution)."""
    n, h, w, c = x.shape
    _check(load().gims_ch_apply(_p(_dev(x, torch.float32)), n, h, w, c, _p(s), _p(b), _p(ah), _p(aw), _p(tau), _p(y), _p(y_split),
                                y_split.stride(0) if y_split is not None else 0, _stream()), "gims_ch_apply")
    return y if y is not None else y_split


def ch_im2col3(x, stride, out, kpad):
    n, h, w, c = x.shape
    _check(load().gims_ch_im2col3(_p(_dev(x, torch.float32)), n, h, w, c, stride, _p(out), out.stride(0), kpad, _stream()), "gims_ch_im2col3")
    return out


def ch_dwconv3(x, wt, bias, y=None, relu6_out=False, res=None, res_scale=1.0, y_split=None):
    n, h, w, c = x.shape
    _check(load().gims_ch_dwconv3(_p(_dev(x, torch.float32)), n, h, w, c, _p(wt), _p(bias), 1 if relu6_out else 0, _p(res), float(res_scale),
                                  _p(y), _p(y_split), y_split.stride(0) if y_split is not None else 0, _stream()), "gims_ch_dwconv3")
    return y if y is not None else y_split


def ch_gate_pw_pw(x, ah, aw, S, z):
    n, h, w, c = x.shape
    _check(load().gims_ch_gate_pw_pw(_p(_dev(x, torch.float32)), n, h, w, c, _p(ah), _p(aw), _p(S["w0"]), _p(S["b0"]), _p(S["w1"]), _p(S["b1"]),
                                     _p(z), _stream()), "gims_ch_gate_pw_pw")
    return z


def ch_input_block(patches, F, tau, out):
    """patches [n, 32, 32, 3] f32 -> FRN + TLU -> SPL32 im2col rows [n*1024, >= 128] of the first convolution."""
    n = patches.shape[0]
    _check(load().gims_ch_input_block(_p(_dev(patches, torch.float32)), n, _p(F["w"]), _p(F["b"]), float(F["eps"]), _p(tau), _p(out), out.stride(0),
                                      _stream()), "gims_ch_input_block")
    return out


def ch_frn_block(x, F, tau, G=None, y=None, y_split=None):
    """FRN (+ CoordAtt gates G) + TLU of one layer, one pass over the activation; F: dict(w, b, eps), G: dict(w1, b1, wh, bh, ww, bw)."""
    n, h, w, c = x.shape
    arr = (C.c_void_p * 6)(*[G[k].data_ptr() for k in ("w1", "b1", "wh", "bh", "ww", "bw")]) if G is not None else None
    _check(load().gims_ch_frn_block(_p(_dev(x, torch.float32)), n, h, c, _p(F["w"]), _p(F["b"]), float(F["eps"]), arr, _p(tau), _p(y), _p(y_split),
                                    y_split.stride(0) if y_split is not None else 0, _stream()), "gims_ch_frn_block")
    return y if y is not None else y_split


def pack_conv3_fragments(w: torch.Tensor) -> torch.Tensor:
    """[cout][cin][3][3] float64/float32 (CPU) -> the bf16 fragment layout of gims_ch_conv_block:
    [step = (ky*3+kx) * cin/16 + ks][nb][plane hi|lo][lane = lh*32 + li][8]."""
    o, i = w.shape[0], w.shape[1]
    assert o % 32 == 0 and i % 16 == 0 and w.shape[2:] == (3, 3)
    x = w.double().permute(2, 3, 1, 0).reshape(9, i // 16, 2, 8, o // 32, 32)         # [tap][ks][lh][e][nb][li]
    x = x.permute(0, 1, 4, 2, 5, 3).reshape(9 * (i // 16), o // 32, 64, 8).float()     # [step][nb][lane][e]
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo], dim=2).contiguous()                                   # [step][nb][plane][lane][8]


def ch_conv_block(xs, n, hin, cin, cout, stride, L, F, tau, G=None, y=None, y_split=None):
    """Fused 3x3 convolution + FRN (+ CoordAtt gates G) + TLU, one workgroup per patch (gims_ch_conv_block).
    xs: SPL32 pixel rows [n*hin*hin, >= 2 cin]; L: dict(wp=packed fragments on the device, b=bias)."""
    arr = (C.c_void_p * 6)(*[G[k].data_ptr() for k in ("w1", "b1", "wh", "bh", "ww", "bw")]) if G is not None else None
    _check(load().gims_ch_conv_block(_p(xs), xs.stride(0), n, hin, cin, cout, stride, _p(L["wp"]), _p(L["b"]), _p(F["w"]), _p(F["b"]), float(F["eps"]), arr,
                                     _p(tau), _p(y), _p(y_split), y_split.stride(0) if y_split is not None else 0, _stream()), "gims_ch_conv_block")
    return y if y is not None else y_split


def ch_conv_block_first(patches, F0, tau0, L, F, tau, G, y_split):
    """Layer 1 in one kernel: patches [n, 32, 32, 3] f32 -> FRN(3) + TLU(3) -> conv 3->32 -> FRN + CoordAtt + TLU -> SPL32 rows."""
    n = patches.shape[0]
    arr = (C.c_void_p * 6)(*[G[k].data_ptr() for k in ("w1", "b1", "wh", "bh", "ww", "bw")]) if G is not None else None
    _check(load().gims_ch_conv_block_first(_p(_dev(patches, torch.float32)), n, _p(F0["w"]), _p(F0["b"]), float(F0["eps"]), _p(tau0), _p(L["wp16"]), _p(L["b"]),
                                           _p(F["w"]), _p(F["b"]), float(F["eps"]), arr, _p(tau), None, _p(y_split), y_split.stride(0), _stream()),
           "gims_ch_conv_block_first")
    return y_split


def ch_sandglass(x, S, out_split):
    """S: dict with the 14 weight tensors of gims_ch_sandglass in order (key 'ptrs': list of tensors)."""
    n, h, w, c = x.shape
    arr = (C.c_void_p * 14)(*[t.data_ptr() for t in S["ptrs"]])
    _check(load().gims_ch_sandglass(_p(_dev(x, torch.float32)), n, h, c, arr, _p(out_split), out_split.stride(0), _stream()), "gims_ch_sandglass")
    return out_split


def ch_l2norm(x, eps, y):
    rows, c = x.shape
    _check(load().gims_ch_l2norm(_p(_dev(x, torch.float32)), rows, c, float(eps), _p(y), _stream()), "gims_ch_l2norm")
    return y


def ch_relu6(x):
    _check(load().gims_ch_relu6(_p(_dev(x, torch.float32)), x.numel(), _stream()), "gims_ch_relu6")
    return x


def train_loss(items, kept0, kept1, gt: torch.Tensor, alpha: float, pos_weight: float, neg_weight: float):
    """forward_train's loss from the solved potentials (gims_train_loss).  items: the per-pair dicts given to make_ot_problems
    (scores, n, m, uv); kept0 / kept1: per pair the int32 device tensors of kept original ids; gt: [K, 3] int64 device tensor
    (b, i0, i1).  Returns (out3 f32 [3] = loss, pos, neg; per-row loss vector [K])."""
    import numpy as np
    dev = gt.device
    assert gt.dtype == torch.int64 and gt.is_contiguous() and (gt.numel() == 0 or gt.shape[1] == 3)
    B = len(items)
    arr = (LossPair * B)()
    for i, (it, k0, k1) in enumerate(zip(items, kept0, kept1)):
        assert k0.dtype == torch.int32 and k1.dtype == torch.int32 and k0.numel() == it["n"] and k1.numel() == it["m"]
        arr[i] = LossPair(_p(it["scores"]), it["scores"].stride(0), it["n"], it["m"], _p(it["uv"]), _p(k0), _p(k1))
    tab = upload(np.frombuffer(bytes(arr), dtype=np.uint8), dev)
    K = int(gt.shape[0])
    loss_vec = torch.empty(max(K, 1), dtype=torch.float32, device=dev)
    tag = torch.empty(max(K, 1) + 2 * B, dtype=torch.int32, device=dev)      # row classes, then 2 group sizes per batch element
    out3 = torch.empty(3, dtype=torch.float32, device=dev)
    _check(load().gims_train_loss(_p(tab), B, _p(gt), K, float(alpha), float(pos_weight), float(neg_weight), _p(loss_vec), _p(tag), _p(out3),
                                  _stream()), "gims_train_loss")
    train_loss.last = dict(table=tab, tag=tag, gt=gt, K=K, B=B)          # what the gradient entry points need again
    return out3, loss_vec[:K]


def sinkhorn_history(items, alpha: float, iters: int):
    """The streamed Sinkhorn solve with the potentials after every iteration recorded (gims_sinkhorn_history): returns the list
    of history buffers; every item's ``uv`` holds the final potentials afterwards (what gims_train_loss reads)."""
    lib = load()
    dev = items[0]["scores"].device
    probs = make_ot_problems(items)
    work = torch.empty(sinkhorn_workspace_bytes(probs), dtype=torch.uint8, device=dev)
    hists = [torch.zeros(int(lib.gims_sinkhorn_history_floats(it["n"], it["m"], iters)), dtype=torch.float32, device=dev) for it in items]
    hp = (C.c_void_p * len(items))(*[h.data_ptr() for h in hists])
    _check(lib.gims_sinkhorn_history(probs, len(items), float(alpha), int(iters), hp, _p(work), work.numel(), _stream()), "gims_sinkhorn_history")
    return hists


def sinkhorn_score_gradients(items, alpha: float, iters: int, pos_weight: float, neg_weight: float, loss_state, hists=None):
    """d loss / d scores (list of [n, m] views with pitch m + 1) and d loss / d bin_score (0-dim tensor) for the loss of
    ``train_loss`` (whose ``last`` state is passed in): a recorded streamed forward solve (``hists`` from sinkhorn_history, made
    here when absent), the loss gradient scattered into zeroed (n+1) x (m+1) buffers (gims_train_loss_grad), the reverse sweep
    through the unrolled iterations (gims_sinkhorn_backward)."""
    lib = load()
    dev = items[0]["scores"].device
    probs = make_ot_problems(items)
    if hists is None:
        hists = sinkhorn_history(items, alpha, iters)
    hp = (C.c_void_p * len(items))(*[h.data_ptr() for h in hists])
    dzs = [torch.zeros((it["n"] + 1, it["m"] + 1), dtype=torch.float32, device=dev) for it in items]
    import numpy as np
    dz_tab = upload(np.asarray([d.data_ptr() for d in dzs], dtype=np.int64), dev)
    st = loss_state
    _check(lib.gims_train_loss_grad(_p(st["table"]), st["B"], _p(st["gt"]), st["K"], float(alpha), _p(st["tag"]), float(pos_weight), float(neg_weight),
                                    _p(dz_tab), _stream()), "gims_train_loss_grad")
    bw = torch.empty(int(lib.gims_sinkhorn_backward_workspace_bytes(probs, len(items))), dtype=torch.uint8, device=dev)
    dalpha = torch.empty(len(items), dtype=torch.float32, device=dev)
    dp = (C.c_void_p * len(items))(*[d.data_ptr() for d in dzs])
    _check(lib.gims_sinkhorn_backward(probs, len(items), float(alpha), int(iters), hp, dp, _p(dalpha), _p(bw), bw.numel(), _stream()),
           "gims_sinkhorn_backward")
    return [d[:it["n"], :it["m"]] for d, it in zip(dzs, items)], dalpha.sum()


def pyramid_layout(h: int, w: int, c: int = 3):
    """(levels [(offset, h, w)], pyramid bytes, scratch bytes) of gims_pyramid_layout."""
    n, pb, sb = C.c_int32(), C.c_size_t(), C.c_size_t()
    _check(load().gims_pyramid_layout(h, w, c, None, 0, C.byref(n), C.byref(pb), C.byref(sb)), "gims_pyramid_layout")
    arr = (PyrLevel * n.value)()
    _check(load().gims_pyramid_layout(h, w, c, arr, n.value, C.byref(n), C.byref(pb), C.byref(sb)), "gims_pyramid_layout")
    return arr, int(pb.value), int(sb.value)


def pyramid_build(img: torch.Tensor):
    """img uint8 [H, W, 3] on the device -> (pyramid buffer uint8, ctypes level table, device level table)."""
    import numpy as np
    assert img.dtype == torch.uint8 and img.is_cuda and img.dim() == 3 and img.is_contiguous()
    h, w, c = img.shape
    levels, pb, sb = pyramid_layout(h, w, c)
    pyr = torch.empty(pb, dtype=torch.uint8, device=img.device)
    scratch = torch.empty(sb, dtype=torch.uint8, device=img.device)
    _check(load().gims_pyramid_build(_p(img), h, w, c, _p(pyr), _p(scratch), _stream()), "gims_pyramid_build")
    dev_levels = upload(np.frombuffer(bytes(levels), dtype=np.uint8), img.device)
    return pyr, levels, dev_levels


def patch_extract(pyr: torch.Tensor, dev_levels: torch.Tensor, n_levels: int, kp4: torch.Tensor, kp_octave: torch.Tensor):
    """kp4 f32 [N, 4] (x, y, size, angle), kp_octave int32 [N] on the device -> (patches f32 [N, 32, 32, 3], bad-count tensor)."""
    assert kp4.dtype == torch.float32 and kp_octave.dtype == torch.int32 and kp4.is_contiguous() and kp4.is_cuda and kp_octave.is_cuda
    n = int(kp4.shape[0])
    out = torch.empty((n, 32, 32, 3), dtype=torch.float32, device=pyr.device)
    bad = torch.empty(1, dtype=torch.int32, device=pyr.device)
    _check(load().gims_patch_extract(_p(pyr), _p(dev_levels), n_levels, _p(kp4), _p(kp_octave), n, _p(out), _p(bad), _stream()), "gims_patch_extract")
    return out, bad

def patch_affine(kp4: torch.Tensor, kp_octave: torch.Tensor):
    """The 2x3 warp matrix (f64 [N, 2, 3]) and pyramid level (int32 [N]) gims_patch_extract derives per keypoint (library.py:96-106)."""
    assert kp4.dtype == torch.float32 and kp_octave.dtype == torch.int32 and kp4.is_contiguous() and kp4.is_cuda and kp_octave.is_cuda
    n = int(kp4.shape[0])
    A = torch.empty((n, 2, 3), dtype=torch.float64, device=kp4.device)
    level = torch.empty(n, dtype=torch.int32, device=kp4.device)
    _check(load().gims_patch_affine(_p(kp4), _p(kp_octave), n, _p(A), _p(level), _stream()), "gims_patch_affine")
    return A, level


# ------------------------------------------------------------------------------------------------ training step
EW_SCALE, EW_ADD, EW_RELU_MASK, EW_RELU, EW_ACC = 0, 1, 2, 3, 4


def _operand(x: torch.Tensor):
    """A 2-D (or batched 3-D) f32 tensor VIEW as a GEMM operand [rows][k]: (transposed flag, pitch, batch stride, rows, k, dims).
    The view must be contiguous along one of its last two dimensions -- `w.t()`, column slices and head slices all qualify.
    (shape and strides are fetched once: this runs ~1 300 times per training step)"""
    sh, sd = x.shape, x.stride()
    nd = len(sh)
    if x.dtype != torch.float32 or not x.is_cuda or nd not in (2, 3):
        raise GimsHipError("gemm operands are 2-D / 3-D f32 device tensors")
    st = sd[0] if nd == 3 else 0
    r, k, s2, s1 = sh[-2], sh[-1], sd[-2], sd[-1]
    if s1 == 1 or k == 1:
        return 0, (s2 if r > 1 else max(k, s2)), st, r, k, nd
    if s2 == 1 or r == 1:
        return 1, (s1 if k > 1 else max(r, s1)), st, r, k, nd
    raise GimsHipError("gemm operand is contiguous along neither of its last two dimensions")


_gemm_work = {}
_gemm_fn = None


GEMM_PRECISION = PREC_BF16X6          # default precision of gemm() outside a gemm_precision() block


class gemm_precision:
    """``with gemm_precision(PREC_BF16X3):`` -- default precision of gemm() for the calling thread inside the block (the training
    step sets it per pass from config['train_precision']); restored on exit, so two models with different settings, or a direct
    gemm() user, never see each other's choice."""

    def __init__(self, precision: int):
        self.precision = int(precision)

    def __enter__(self):
        self._prev = getattr(_TLS, "gemm_prec", None)
        _TLS.gemm_prec = self.precision
        return self

    def __exit__(self, *exc):
        _TLS.gemm_prec = self._prev
        return False


def gemm(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor | None = None, *, alpha=1.0, beta=0.0, bias=None, residual=None, act=ACT_NONE,
         precision=None):
    """out = alpha * a @ b^T + beta * out (+ bias over the last dimension) (+ residual), then act (gims_gemm_f32, split-bf16 MFMA,
    f32 class by default).  a: [.., m, k], b: [.., n, k] views (either may be a transposed view), out: [.., m, n] with unit
    stride along n.  Batched when the tensors are 3-D."""
    global _gemm_fn
    ta, lda, sa, m, k, nda = _operand(a)
    tb, ldb, sb, n, kb, ndb = _operand(b)
    if kb != k or nda != ndb:
        raise GimsHipError(f"gemm: shapes {tuple(a.shape)} x {tuple(b.shape)}^T do not agree")
    batch = a.shape[0] if nda == 3 else 1
    if out is None:
        out = torch.empty((batch, m, n) if nda == 3 else (m, n), dtype=torch.float32, device=a.device)
    osh, osd = out.shape, out.stride()
    if osh[-2] != m or osh[-1] != n or (osd[-1] != 1 and n > 1) or out.dtype != torch.float32:
        raise GimsHipError("gemm: bad output tensor")
    ldr = sr = 0
    if residual is not None:
        rsd = residual.stride()
        ldr, sr = rsd[-2], (rsd[0] if len(rsd) == 3 else 0)
    g = Gemm(a.data_ptr(), b.data_ptr(), out.data_ptr(), _p(bias), _p(residual), lda, ldb, osd[-2] if m > 1 else max(n, osd[-2]), ldr, sa, sb,
             osd[0] if len(osd) == 3 else 0, sr, m, n, k, batch, ta, tb, int(act), 0, float(alpha), float(beta))
    if precision is None:
        precision = getattr(_TLS, "gemm_prec", None)
    g.precision = GEMM_PRECISION if precision is None else int(precision)
    stream = _stream()
    if k >= 512:                                  # split-K workspace (one arena per device and stream; stream order protects it)
        key = (a.device, stream)
        w = _gemm_work.get(key)
        if w is None:
            w = _gemm_work[key] = torch.empty(32 << 20, dtype=torch.float32, device=a.device)
        g.work, g.work_floats = w.data_ptr(), w.numel()
    if _gemm_fn is None:
        _gemm_fn = load().gims_gemm_f32
    rc = _gemm_fn(C.byref(g), stream)
    if rc != 0:
        _check(rc, "gims_gemm_f32")
    return out


def segments(ranges) -> Segments:
    """ranges: [(first row, rows)] -- the rows every call of a module covers in the reference (image 0 of the batch, image 1)."""
    sg = Segments()
    sg.n = len(ranges)
    for i, (o, r) in enumerate(ranges):
        sg.off[i], sg.rows[i] = int(o), int(r)
    return sg


def batchnorm_train_forward(x, sg: Segments, gamma, beta, eps, momentum, running_mean, running_var, relu: bool, y=None):
    """nn.BatchNorm1d in train() mode (+ ReLU) on x [rows, c]; returns (y, save).  Running statistics are updated in place."""
    c = x.shape[1]
    y = torch.empty_like(x) if y is None else y
    save = torch.empty((sg.n, c, 2), dtype=torch.float32, device=x.device)
    work = torch.empty(int(load().gims_batchnorm_workspace_floats(C.byref(sg), c)), dtype=torch.float32, device=x.device)
    _check(load().gims_batchnorm_train_forward(_p(x), x.stride(0), c, C.byref(sg), _p(gamma), _p(beta), float(eps), float(momentum), _p(running_mean),
                                               _p(running_var), _p(save), _p(y), y.stride(0), int(relu), _p(work), _stream()), "gims_batchnorm_train_forward")
    return y, save


def batchnorm_train_backward(x, dy, sg: Segments, save, gamma, beta, relu: bool, dx=None):
    """Returns (dx, dgamma, dbeta) for the module of batchnorm_train_forward; dy: gradient of its (post-ReLU) output."""
    c = x.shape[1]
    dx = torch.empty_like(x) if dx is None else dx
    dg = torch.empty(c, dtype=torch.float32, device=x.device)
    db = torch.empty(c, dtype=torch.float32, device=x.device)
    work = torch.empty(int(load().gims_batchnorm_workspace_floats(C.byref(sg), c)), dtype=torch.float32, device=x.device)
    _check(load().gims_batchnorm_train_backward(_p(x), x.stride(0), _p(dy), dy.stride(0), c, C.byref(sg), _p(save), _p(gamma), _p(beta), int(relu),
                                                _p(dx), dx.stride(0), _p(dg), _p(db), _p(work), _stream()), "gims_batchnorm_train_backward")
    return dx, dg, db


def softmax_rows_(s: torch.Tensor, cols: int):
    """In-place softmax over the first `cols` entries of every row of s [batch, rows, ld]."""
    assert s.dim() == 3 and s.stride(2) == 1
    _check(load().gims_softmax_rows(_p(s), s.stride(1), s.shape[1], int(cols), s.shape[0], s.stride(0), _stream()), "gims_softmax_rows")
    return s


def softmax_rows_backward_(prob: torch.Tensor, dp: torch.Tensor, cols: int):
    assert prob.shape == dp.shape and prob.stride() == dp.stride() and prob.dim() == 3
    _check(load().gims_softmax_rows_backward(_p(prob), _p(dp), prob.stride(1), prob.shape[1], int(cols), prob.shape[0], prob.stride(0), _stream()),
           "gims_softmax_rows_backward")
    return dp


_tattn_work = {}


def train_attn_problems(problems):
    """[(q_off, nq, k_off, nk)] -> (ctypes array, n): problem i attends the query rows [q_off, q_off + nq) to the source rows [k_off, k_off + nk)."""
    arr = (TrainAttnProblem * len(problems))()
    for i, (qo, nq, ko, nk) in enumerate(problems):
        arr[i].q_off, arr[i].nq, arr[i].k_off, arr[i].nk = int(qo), int(nq), int(ko), int(nk)
    return arr, len(problems)


TRAIN_ATTN_REVERSE_F32, TRAIN_ATTN_REVERSE_BF16X3 = 0, 1


def _train_attn_args(qkv, problems, heads, o, lse, d_o=None, d_qkv=None, reverse_precision=1):
    rows, d = qkv.shape[0], qkv.shape[1] // 3
    assert qkv.dtype == torch.float32 and qkv.stride(1) == 1 and o.stride(1) == 1 and lse.is_contiguous() and lse.shape == (heads, rows)
    need = int(load().gims_train_attention_workspace_floats(rows, heads))
    key = (qkv.device, _stream())
    w = _tattn_work.get(key)
    if w is None or w.numel() < need:
        if w is not None:
            _colsum_retired.append(w)                 # (see colsum: a launch on a non-torch stream may still read the outgrown buffer)
        w = _tattn_work[key] = torch.empty(need, dtype=torch.float32, device=qkv.device)
    arr, n = problems if isinstance(problems, tuple) else train_attn_problems(problems)
    g = TrainAttnArgs(qkv.data_ptr(), qkv.stride(0), rows, d, heads, 1.0 / math.sqrt(d // heads), n, arr, o.data_ptr(), o.stride(0), lse.data_ptr(),
                      _p(d_o), d_o.stride(0) if d_o is not None else 0, _p(d_qkv), d_qkv.stride(0) if d_qkv is not None else 0, w.data_ptr(), w.numel(),
                      int(reverse_precision))
    return g, arr


def train_attention_forward(qkv: torch.Tensor, problems, heads: int, o: torch.Tensor | None = None, lse: torch.Tensor | None = None):
    """o = softmax(Q K^T / sqrt(64)) V for every problem and head of one layer (qkv [rows, 3 d]: packed Q | K | V, heads contiguous), and the
    row statistic lse [heads, rows] the reverse pass recomputes the probabilities from.  Returns (o, lse)."""
    rows, d = qkv.shape[0], qkv.shape[1] // 3
    o = torch.empty((rows, d), dtype=torch.float32, device=qkv.device) if o is None else o
    lse = torch.empty((heads, rows), dtype=torch.float32, device=qkv.device) if lse is None else lse
    g, keep = _train_attn_args(qkv, problems, heads, o, lse)
    _check(load().gims_train_attention_forward(C.byref(g), _stream()), "gims_train_attention_forward")
    return o, lse


def train_attention_backward(qkv, o, lse, d_o, problems, heads: int, d_qkv: torch.Tensor | None = None, precision=None):
    """d_qkv [rows, 3 d] from the gradient d_o of train_attention_forward's output.  precision: PREC_BF16X3 (three bf16 passes; also the default
    inside a gemm_precision(PREC_BF16X3) block, as the training step's reverse pass is) or anything else = exact f32 products."""
    if precision is None:
        precision = getattr(_TLS, "gemm_prec", None)
    rp = TRAIN_ATTN_REVERSE_BF16X3 if precision == PREC_BF16X3 else TRAIN_ATTN_REVERSE_F32
    d_qkv = torch.empty_like(qkv) if d_qkv is None else d_qkv
    assert d_o.stride(1) == 1 and d_qkv.stride(1) == 1
    g, keep = _train_attn_args(qkv, problems, heads, o, lse, d_o, d_qkv, rp)
    _check(load().gims_train_attention_backward(C.byref(g), _stream()), "gims_train_attention_backward")
    return d_qkv


_colsum_work = {}
_colsum_retired = []


def colsum(x: torch.Tensor, out: torch.Tensor | None = None, beta=0.0):
    """out[c] = beta * out[c] + sum_rows x[row, c] (fixed summation order)."""
    rows, c = x.shape
    if out is None:
        out = torch.empty(c, dtype=torch.float32, device=x.device)
    need = int(load().gims_colsum_workspace_floats(rows, c))
    key = (x.device, _stream())
    w = _colsum_work.get(key)
    if w is None or w.numel() < need:
        if w is not None:
            # a kernel enqueued on this (possibly non-torch) stream may still be reading the outgrown workspace, and torch's allocator orders
            # re-use only against ITS streams: the old buffer is kept alive for the life of the process (regrowth is rare, the buffers small)
            _colsum_retired.append(w)
        w = _colsum_work[key] = torch.zeros(max(need, 1 << 16), dtype=torch.float32, device=x.device)
        if getattr(_TLS, "pin", None) is not None:
            torch.cuda.current_stream(x.device).synchronize()      # the zero fill ran on torch's stream, the kernel may run on another (on_stream)
    # (the kernel leaves its counters -- the first 64 words -- zeroed; partials are overwritten before they are read)
    _check(load().gims_colsum(_p(x), x.stride(0), rows, c, float(beta), _p(out), _p(w), _stream()), "gims_colsum")
    return out


def elementwise(op: int, out, a, b=None, alpha=1.0):
    rows, cols = a.shape
    _check(load().gims_elementwise(int(op), _p(out), out.stride(0), _p(a), a.stride(0), _p(b), (b.stride(0) if b is not None else 0), rows, cols,
                                   float(alpha), _stream()), "gims_elementwise")
    return out


def permute3(dst, src, shape, dstrides, sstrides, accumulate=False):
    _check(load().gims_permute3(_p(dst), _p(src), *[int(v) for v in shape], *[int(v) for v in dstrides], *[int(v) for v in sstrides], int(accumulate),
                                _stream()), "gims_permute3")
    return dst


def sage_mean_transposed(g, indptr, indices, out=None):
    n, c = g.shape
    out = torch.empty_like(g) if out is None else out
    _check(load().gims_sage_mean_transposed(_p(g), g.stride(0), _p(indptr), _p(indices), n, c, _p(out), out.stride(0), _stream()),
           "gims_sage_mean_transposed")
    return out


def normalize_keypoints(kpts, norm3, seg):
    out = torch.empty_like(kpts)
    _check(load().gims_normalize_keypoints(_p(kpts), _p(norm3), _p(seg), kpts.shape[0], _p(out), _stream()), "gims_normalize_keypoints")
    return out


def layernorm_backward(x, dy, a2, b2, relu: bool, eps=1e-6):
    """Reverse pass of layernorm_act on x [rows, c]: returns (dx, d a_2, d b_2)."""
    rows, c = x.shape
    dx = torch.empty_like(x)
    gb = torch.empty((rows, c), dtype=torch.float32, device=x.device)
    ga = torch.empty((rows, c), dtype=torch.float32, device=x.device)
    _check(load().gims_layernorm_backward(_p(x), x.stride(0), _p(dy), dy.stride(0), rows, c, _p(a2), _p(b2), float(eps), int(relu), _p(dx), dx.stride(0),
                                          _p(gb), _p(ga), _stream()), "gims_layernorm_backward")
    return dx, colsum(ga), colsum(gb)


def head_pack(proj_w, proj_b, merge_w, wqkv, bqkv, wm, heads: int, to_params: bool):
    """All head-interleave permutations of one attention layer in one launch (gims_head_pack): parameters -> packed tensors, or
    packed gradients -> parameter-layout gradients."""
    pw = (C.c_void_p * 3)(*[t.data_ptr() for t in proj_w])
    pb = (C.c_void_p * 3)(*[t.data_ptr() for t in proj_b])
    _check(load().gims_head_pack(pw, pb, _p(merge_w), _p(wqkv), _p(bqkv), _p(wm), merge_w.shape[0], int(heads), int(to_params), _stream()), "gims_head_pack")


ADAM_TENSOR_DTYPE = [("param", "<u8"), ("grad", "<u8"), ("exp_avg", "<u8"), ("exp_avg_sq", "<u8"), ("n", "<i8"), ("group", "<i4"), ("reserved", "<i4")]


def adam_step(table, groups):
    """One fused Adam step (gims_adam_step).  table: C-contiguous NumPy structured array of dtype ADAM_TENSOR_DTYPE (= gims_adam_tensor:
    device pointers of contiguous float32 tensors, element count, group index); groups: at most 8 dicts with lr, beta1, beta2, eps,
    weight_decay, step (1-based, after the increment)."""
    import numpy as np
    assert table.dtype == np.dtype(ADAM_TENSOR_DTYPE) and table.flags["C_CONTIGUOUS"] and table.dtype.itemsize == C.sizeof(AdamTensor)
    gt = (AdamGroup * max(len(groups), 1))()
    for i, g in enumerate(groups):
        gt[i].lr, gt[i].beta1, gt[i].beta2, gt[i].eps, gt[i].weight_decay, gt[i].step = (float(g["lr"]), float(g["beta1"]), float(g["beta2"]), float(g["eps"]),
                                                                                      float(g["weight_decay"]), int(g["step"]))
    _check(load().gims_adam_step(table.ctypes.data_as(C.POINTER(AdamTensor)), len(table), gt, len(groups), _stream()), "gims_adam_step")

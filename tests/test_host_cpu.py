"""CPU-only tests: the C-ABI library loads and exports every symbol include/gims_hip.h declares, the host
mirror keeps the reference's interface (state-dict names, config keys, checkpoint layouts, error behaviour).
No compute call is made here (there is no GPU in the build container)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from gims_amd import GMatcher, Matching, hip, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gims_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gims_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    assert declared == set(hip.EXPORTS), declared ^ set(hip.EXPORTS)
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert hip.load().gims_abi_version() == hip.ABI_VERSION == 2


def test_struct_layouts_match_header(tmp_path):
    """Sizes and field offsets of the ctypes mirrors against what a C compiler makes of include/gims_hip.h (gcc, LP64)."""
    import subprocess
    pairs = [("gims_linear_args", hip.LinearArgs, ["a0", "w", "bias", "out_f32", "m", "act", "scale", "a0_lo", "out_hi", "ld_split", "flags", "conv_h", "guard", "range_stat"]),
             ("gims_attn_guard", hip.AttnGuard, ["stat", "mean_thr", "range_limit", "n_heads", "kind", "max_thr"]),
             ("gims_attn_args", hip.AttnArgs, ["qkv", "q_col", "problems", "n_heads", "out", "ld_split", "flags", "stat", "guard"]),
             ("gims_train_attn_problem", hip.TrainAttnProblem, ["nk"]),
             ("gims_train_attn_args", hip.TrainAttnArgs, ["qkv", "rows", "d", "scale", "problems", "o", "lse", "d_o", "d_qkv", "work", "work_floats", "reverse_precision"]),
             ("gims_ot_problem", hip.OtProblem, []), ("gims_agc_image", hip.AgcImage, ["kept", "max_edges_dir", "info"]),
             ("gims_pack_image", hip.PackImage, []), ("gims_ingest_image", hip.IngestImage, []), ("gims_op", hip.Op, ["u"]),
             ("gims_aux_args", hip.AuxArgs, ["fn", "p", "i"])]
    body = "".join('printf("%s %%zu\\n", sizeof(%s));\n' % (c, c) + "".join('printf("%s.%s %%zu\\n", offsetof(%s, %s));\n' % (c, f, c, f) for f in fs)
                   for c, _, fs in pairs)
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gims_hip.h"\nint main(void) {\n' + body + "return 0; }\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for c, py, fs in pairs:
        assert int(got[c]) == ctypes.sizeof(py), (c, got[c], ctypes.sizeof(py))
        for f in fs:
            assert int(got[f"{c}.{f}"]) == getattr(py, f).offset, (c, f)


def test_pyramid_layout_equals_the_reference_schedule(golden_dir):
    """gims_pyramid_layout (host arithmetic only): number of levels and every level's size, against what the reference's
    buildGaussianPyramid produced for the same image sizes (tests/golden/patch_pyramid_calls.npz, tools/gen_golden_patches.py)."""
    import numpy as np
    g = np.load(os.path.join(golden_dir, "patch_pyramid_calls.npz"))
    for (h, w) in g["shapes"]:
        levels, _, _ = hip.pyramid_layout(int(h), int(w), 3)
        assert len(levels) == int(g[f"{h}x{w}/n_levels"])
        assert [[L.h, L.w] for L in levels] == g[f"{h}x{w}/level_shapes"].tolist()


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(hip.GimsHipError):
        hip.load(str(tmp_path / "nope.so"))


def test_state_dict_names_and_checkpoint_layouts(tmp_path, synth_sd):
    m = GMatcher({})
    names = list(m.state_dict().keys())
    assert len(names) == 348 and "bin_score" in names
    assert "gnn.layers.17.attn.proj.2.weight" in names and "kenc.encoder.12.bias" in names
    assert "gnn_encoder.layers.0.fc_neigh.weight" in names and "gnn.layers.0.mlp.1.running_var" in names
    assert set(names) == set(synth_sd.keys())
    sd_t = {k: torch.from_numpy(np.asarray(v)) for k, v in synth_sd.items()}
    for layout in ("raw", "model", "ema", "module."):
        blob = {"raw": sd_t, "model": {"model": sd_t, "ema": None}, "ema": {"ema": sd_t, "model": {}},
                "module.": {"module." + k: v for k, v in sd_t.items()}}[layout]
        p = tmp_path / f"{layout}.pt"
        torch.save(blob, p)
        mm = GMatcher({"weights_path": str(p)})
        assert torch.equal(mm.state_dict()["final_proj.weight"], sd_t["final_proj.weight"])
    # older-DGL SAGEConv bias layout
    old = dict(synth.make_state_dict(123, sage_bias_layout="bias"))
    assert "gnn_encoder.layers.0.bias" in old
    mm = GMatcher({})
    mm.load_state_dict(old)
    assert torch.equal(mm.state_dict()["gnn_encoder.layers.1.fc_self.bias"], torch.from_numpy(old["gnn_encoder.layers.1.bias"]))


def test_default_config_keys_match_reference():
    for k, v in {"descriptor_dim": 256, "weights_path": None, "keypoint_encoder": [32, 64, 128, 256],
                 "transformer_layers": ["self", "cross"] * 9, "sinkhorn_iterations": 100, "match_threshold": 0.2,
                 "use_layernorm": False, "input_dim": 256, "num_heads": 4}.items():
        assert GMatcher.default_config[k] == v


def test_no_cpu_fallback_and_matching_shell():
    m = Matching({})
    pair = synth.make_pair(64, 1000)
    data = {k: torch.from_numpy(v) for k, v in pair.items() if k != "gt_perm"}
    with pytest.raises(hip.GimsHipError):
        m(data)
    with pytest.raises(NotImplementedError):
        m({"image0": pair["image0"], "image1": pair["image1"]})


def test_matching_front_end_hook_is_called_like_sift_forward():
    """Without keypoints, Matching calls its front end once per image with the reference's argument dict
    (models/matching.py:17-24) and feeds the stacked lists to GMatcher (which then refuses CPU tensors: no fallback)."""
    pair = synth.make_pair(64, 1000)
    calls = []

    def fake_front_end(d, device):
        calls.append((d["image"].shape, d["max_keypoints"], d["carhynet"], device))
        s = "0" if len(calls) == 1 else "1"
        return {"keypoints": [torch.from_numpy(pair["keypoints" + s][0])], "scores": [torch.from_numpy(pair["scores" + s][0])],
                "descriptors": [torch.from_numpy(pair["descriptors" + s][0])]}

    m = Matching({"front_end": fake_front_end, "max_keypoints": 77})
    with pytest.raises(hip.GimsHipError):
        m({"image0": pair["image0"], "image1": pair["image1"], "carhynet": "net", "device": "cpu"})
    assert calls == [(pair["image0"].shape, 77, "net", "cpu"), (pair["image1"].shape, 77, "net", "cpu")]
    assert "front_end" not in m.gmodel.config
    with pytest.raises(TypeError):
        Matching({"front_end": 3})


def test_product_code_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "gims_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f


def test_pose_auc_and_summarize_equal_reference():
    """gims_amd.evalh.pose_auc against the reference's pose_auc outputs (tests/golden/eval_metrics.npz)."""
    from gims_amd import evalh
    from tests.helpers import load_golden
    g = load_golden("eval_metrics")
    for i in range(4):
        np.testing.assert_allclose(evalh.pose_auc(g[f"errors{i}"], [5, 10, 25]), g[f"auc{i}"], rtol=0, atol=1e-12)
    rec = np.zeros((3, 16))
    rec[:, 0] = [100, 5, 50]          # n_valid: the second pair has too few matches
    rec[:, 9] = rec[:, 10] = 1
    rec[:, 4], rec[:, 5], rec[:, 7], rec[:, 8] = [0.5, 0.1, 1.0], [0.25, 0.1, 0.75], [2.0, 1.0, 30.0], [1.0, 1.0, 3.0]
    s = evalh.summarize(rec)
    assert s["n_pairs"] == 2 and s["precision"] == 75.0 and s["recall"] == 50.0
    np.testing.assert_allclose(s["auc_ransac"], [100 * a for a in evalh.pose_auc([1.0, 3.0])], atol=1e-12)


def test_training_front_end_pads_to_max_keypoints():
    """sift_forward's is_train branch (utils/common.py:866-880): random extra size-1 keypoints up to max_keypoints, drawn with the
    reference's sequence of np.random calls."""
    from gims_amd import frontend
    kps = [frontend.PaddedKeyPoint(3.0, 4.0)]
    np.random.seed(5)
    out = frontend.pad_training_keypoints(kps, 6, (480, 640, 3))
    np.random.seed(5)
    c = np.random.random((5, 2)) * 640
    c[:, 1] = np.random.random(5) * 480
    assert len(out) == 6 and out[0] is kps[0]
    assert [k.pt for k in out[1:]] == [(float(x), float(y)) for x, y in c]
    assert all(k.size == 1.0 and k.octave == 0 and k.response == 0.0 for k in out[1:])
    assert frontend.pad_training_keypoints(kps * 7, 6, (480, 640, 3)) == kps * 7          # nothing to add
    kp4, octv, resp = frontend.keypoint_arrays(out)
    assert kp4.shape == (6, 4) and octv.tolist() == [0] * 6


def test_training_entry_points_validate_arguments_without_a_gpu():
    """Argument checks of the training-step ABI run before any HIP call: bad arguments come back as GIMS_EINVAL (-1) with a message,
    also on a machine without a GPU."""
    import ctypes as C
    lib = hip.load()
    g = hip.Gemm()                                        # all null
    assert lib.gims_gemm_f32(C.byref(g), None) == -1 and b"gims_gemm_f32" in lib.gims_last_error()
    assert lib.gims_gemm_f32(None, None) == -1
    sg = hip.Segments()
    sg.n = 0
    assert lib.gims_batchnorm_workspace_floats(C.byref(sg), 32) == 0
    assert lib.gims_batchnorm_train_forward(None, 0, 32, C.byref(sg), None, None, 1e-5, 0.1, None, None, None, None, 0, 1, None, None) == -1
    assert lib.gims_softmax_rows(None, 0, 4, 4, 1, 0, None) == -1
    assert lib.gims_colsum(None, 0, 4, 4, 0.0, None, None, None) == -1
    assert lib.gims_colsum_workspace_floats(1000, 512) == 64 + 8 * 512
    assert lib.gims_layernorm_backward(None, 0, None, 0, 4, 1, None, None, 1e-6, 1, None, 0, None, None, None) == -1
    assert lib.gims_head_pack(None, None, None, None, None, None, 256, 4, 0, None) == -1
    assert lib.gims_permute3(None, None, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, None) == -1
    assert lib.gims_sinkhorn_history_floats(10, 12, 5) == 6 * 24


def test_adam_table_layout_and_cpu_refusal():
    """gims_adam_tensor / gims_adam_group as the binding lays them out (include/gims_hip.h), and the optimizer's refusal to step CPU
    parameters (there is no CPU path)."""
    import ctypes as C
    import numpy as np
    import pytest
    import torch
    from gims_amd import hip
    from gims_amd.optim import Adam
    assert C.sizeof(hip.AdamTensor) == 48 == np.dtype(hip.ADAM_TENSOR_DTYPE).itemsize
    assert [f[0] for f in hip.AdamTensor._fields_] == [f[0] for f in hip.ADAM_TENSOR_DTYPE]
    assert C.sizeof(hip.AdamGroup) == 48
    p = [torch.nn.Parameter(torch.zeros(4))]
    with pytest.raises(NotImplementedError):
        Adam(p, amsgrad=True)
    with pytest.raises(ValueError):
        Adam(p, lr=-1.0)
    o = Adam(p, lr=1e-3, weight_decay=1e-4)
    assert o.param_groups[0]['betas'] == (0.9, 0.999) and o.param_groups[0]['weight_decay'] == 1e-4
    o.add_param_group({'params': [torch.nn.Parameter(torch.zeros(2))], 'weight_decay': 0.5})       # train.py:56
    assert len(o.param_groups) == 2 and o.param_groups[1]['lr'] == 1e-3
    p[0].grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        o.step()


def test_hot_kernels_use_no_scratch():
    """Round 3 found the staging registers of several attention kernels living in scratch (a global-memory round trip per key tile; the split-bf16
    kernel went 1.26 -> 0.73 ms per launch once K and V took the LDS-DMA path instead).  This pins the invariant: the kernels of the hot path that are
    meant to be scratch-free compile to `ScratchSize: 0` for gfx950 (hipcc cross-compiles here; the remark comes from the register allocator)."""
    import os
    import re
    import shutil
    import subprocess
    import tempfile
    from gims_amd import build as B
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        import pytest
        pytest.skip("no hipcc")
    must_be_clean = {"attention.hip": ["attention8_bf16_kernel", "attention_bf16_kernel", "attention_x3_kernel", "attention_split_kernelILi2"],
                     "linear6.hip": ["linear_x6_kernel"], "linear.hip": ["linear_x3p_kernel"],
                     "carhynet.hip": ["ch_conv_block_kernel", "ch_sandglass_kernelILi32"]}
    with tempfile.TemporaryDirectory() as tmp:
        for src, names in must_be_clean.items():
            cmd = [hipcc, *B.FLAGS, *B.EXTRA.get(src, []), "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(B.CSRC, src), "-o", os.path.join(tmp, "x.o")]
            out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True).stdout
            cur, scratch = None, {}
            for line in out.splitlines():
                m = re.search(r"Function Name: (\S+)", line)
                if m:
                    cur = m.group(1)
                m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
                if m and cur:
                    scratch[cur] = int(m.group(1))
            for n in names:
                hits = {k: v for k, v in scratch.items() if n in k}
                assert hits, f"{src}: no kernel named like {n} in the compiler remarks"
                assert all(v == 0 for v in hits.values()), f"{src}: scratch in a hot kernel: {hits}"


def test_auto_attention_routes_on_the_tail_and_the_range():
    """The decision rule of 'auto' on hand-made statistics (no GPU work: the folding of a read-back into per-layer modes):
    a head with 5 % one-hot rows among diffuse ones has a small MEAN row maximum (0.05) but a tail fraction of 0.05 -> half;
    operands beyond `attention_f16_range` -> split-bf16; layers only move up."""
    m = GMatcher({}).eval()
    L, H = m.n_layers, 4
    st = m.__dict__["_attn_auto"] = dict(gen=7, mode=[2] * L, calibrated=False, peak=np.zeros((L, H)), peak_max=np.zeros((L, H)),
                                         tail=np.zeros((L, H)), range=np.zeros((L, 3)), switched=[], batches={}, redone=np.zeros(L, dtype=np.int64))

    class Ev:
        def synchronize(self):
            pass

    def feed(mean, tail, rng):
        raw = np.zeros((L, H + 1, 4), dtype=np.int64)
        raw[:, :H, 1] = 1000
        raw[:, :H, 0] = np.round(np.asarray(mean) * 1000 * 2 ** 24).astype(np.int64)
        raw[:, :H, 3] = np.round(np.asarray(tail) * 1000).astype(np.int64)
        raw[:, H, :3] = np.asarray(rng, dtype=np.float32).view(np.uint32).astype(np.int64)
        m.__dict__["_attn_pending"] = {0: [torch.from_numpy(raw), Ev(), 7]}
        m._attention_stats_consume()
        return m.attention_report()["modes"]

    mean, tail, rng = np.full((L, H), 0.01), np.zeros((L, H)), np.full((L, 3), 20.0)
    mean[1, 2] = 0.3                     # plainly peaked head
    mean[2, 0], tail[2, 0] = 0.05, 0.05  # small mean, fat tail
    mean[3, 1], rng[3, 1] = 0.5, 5.0e4   # peaked AND out of half's range
    rng[4, 0] = 5.0e4                    # wide operands but diffuse: bf16 has the range
    modes = feed(mean, tail, rng)
    assert modes[:5] == ["bf16", "f16", "f16", "bf16x3", "bf16"] and set(modes[5:]) == {"bf16"}
    mean2 = np.full((L, H), 0.01)        # a later, calmer measurement never moves a layer down ...
    assert feed(mean2, np.zeros((L, H)), np.full((L, 3), 20.0)) == modes
    rng3 = np.full((L, 3), 20.0); rng3[1, 2] = 4.0e4      # ... and a half layer whose operands grow goes up
    mean3 = mean2.copy(); mean3[1, 2] = 0.3
    modes3 = feed(mean3, np.zeros((L, H)), rng3)
    assert modes3[1] == "bf16x3" and m.attention_report()["switched"] == [1]
    # the device's record of a redo (stat[H][3], set by a guarded attention launch that fired) is counted per layer
    raw = np.zeros((L, H + 1, 4), dtype=np.int64)
    raw[5, H, 3] = 1
    m.__dict__["_attn_pending"] = {0: [torch.from_numpy(raw), Ev(), 7]}
    assert m.attention_report()["redone"].tolist() == [0] * 5 + [1] + [0] * (L - 6)
    # round 6: ONE sharply peaked row inside a diffuse bf16 layer (the head's largest row maximum reaches attention_auto_rowmax = 0.5 while mean
    # and tail stay under their thresholds) is recorded as `rare` and makes forward() repeat the batch with the device-side guards -- the layer is
    # NOT moved up; the same maximum on a layer that is not on the bf16 tier means nothing
    raw = np.zeros((L, H + 1, 4), dtype=np.int64)
    raw[:, :H, 1] = 1000
    raw[:, :H, 0] = int(0.01 * 1000 * 2 ** 24)
    raw[:, H, :3] = np.asarray(np.full(3, 20.0), dtype=np.float32).view(np.uint32).astype(np.int64)
    raw[7, 3, 2] = int(0.97 * 2 ** 24)      # layer 7 (bf16 tier): a row at 0.97
    raw[1, 0, 2] = int(0.99 * 2 ** 24)      # layer 1 (split-bf16 by now): irrelevant
    before = m.attention_report()["modes"]
    m.__dict__["_attn_pending"] = {0: [torch.from_numpy(raw), Ev(), 7]}
    assert m._attention_stats_consume() == 0
    rep = m.attention_report()
    assert rep["modes"] == before and rep["rare"].tolist() == [0] * 7 + [1] + [0] * (L - 8) and m._attn_auto["rare_last"] is True
    raw[7, 3, 2] = int(0.3 * 2 ** 24)
    m.__dict__["_attn_pending"] = {0: [torch.from_numpy(raw), Ev(), 7]}
    m._attention_stats_consume()
    assert m._attn_auto["rare_last"] is False and m.attention_report()["rare"].sum() == 1
    # ... but a layer that does it on attention_auto_rare_batches (3) batches is no outlier: it moves to the half tier (cheaper than a redo per batch)
    raw[7, 3, 2] = int(0.9 * 2 ** 24)
    for k in (2, 3):
        m.__dict__["_attn_pending"] = {0: [torch.from_numpy(raw), Ev(), 7]}
        moved = m._attention_stats_consume()
        assert m.attention_report()["rare"][7] == k and moved == (1 if k == 3 else 0)
    assert m.attention_report()["modes"][7] == "f16" and m.attention_report()["modes"][8] == "bf16"


def test_graph_build_flag_words_decide_the_repeat():
    """info[7] of a graph build (include/gims_hip.h): bit 1 = the predicted percentile window was missed -- every output of that image is void,
    its overflow bit too, so the repeat with the robust flow comes before any capacity growth; bit 0 alone = grow the edge buffers; a robust
    build that reports a miss is an error (it predicts nothing)."""
    import numpy as np
    import pytest
    from gims_amd import GMatcher, hip
    R = GMatcher._agc_retry
    assert R(np.array([0, 0, 0]), False) is None
    assert R(np.array([0, 1, 0]), False) == "grow"
    assert R(np.array([0, 2, 0]), False) == "robust"
    assert R(np.array([1, 3, 0]), False) == "robust"          # a missed image may also claim an overflow: void
    assert R(np.array([1, 0, 0]), True) == "grow"
    with pytest.raises(hip.GimsHipError):
        R(np.array([0, 2]), True)


def test_bench_line_is_compact_and_parses_from_the_stored_tail():
    """The driver keeps the last 8000 characters of stdout and parses the last line: bench.py's line must stay under 4 KB
    whatever the full record holds (round 4 lost its measurement to a 24-KB line), also with 8 ranks and every extra block."""
    import json
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full = json.load(open(os.path.join(root, "profiles", "r04_bench_default.json")))        # a real 24-KB record
    assert len(json.dumps(full)) > 20000
    full["n_gpus"], full["world_size_seen"] = 8, 8
    full["ranks"] = [{"rank": r, "device": r, "device_name": "AMD Instinct MI355X", "backend": "nccl", "pid": 1000 + r} for r in range(8)]
    full["also"]["readme_boat_15k"] = dict(full["also"]["2x4096_eval_setting"], agc_ms_per_image=3.21)
    line = bench.compact_line(full)
    assert "\n" not in line and len(line) < bench.LINE_LIMIT <= 4096
    stdout = "x" * 20000 + "\n" + line + "\n"
    got = json.loads(stdout[-8000:].strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "cross_attention", "world_size_seen", "ranks"):
        assert k in got, k
    assert set(got["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"}
    assert set(got["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert got["config"]["workload"] and got["also"]["2x1024"]["value"] > 0
    assert abs(got["value"] - full["value"]) / full["value"] < 1e-3
    # a record that is still too long sheds its optional parts instead of the contract's keys
    full["config"]["workload"] = "w" * 3000
    line = bench.compact_line(full)
    assert len(line) < bench.LINE_LIMIT and "roofline" in json.loads(line) and "cpu_baseline" in json.loads(line)


def test_bench_ranks_keep_to_disjoint_core_slices():
    """bench.py N > 1: every rank restricts itself to its own slice of the inherited CPU set before it touches the GPU (the step is host-bound within
    ~1 % of the GPU time: N interpreters must not share cores).  Equal contiguous shares, disjoint, inside the inherited set, whatever that set
    looks like (a cgroup may hand out a sparse one); too few cores -> the affinity is left alone."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for cores, world in ((range(256), 8), (range(0, 128, 2), 4), ([3, 4, 5, 9, 10, 11, 12, 40], 2), (range(96), 8)):
        cores = list(cores)
        parts = [bench.rank_affinity(cores, r, world) for r in range(world)]
        assert all(p is not None and len(p) == len(cores) // world for p in parts)
        flat = [c for p in parts for c in p]
        assert len(set(flat)) == len(flat) and set(flat) <= set(cores)
        assert all(p == sorted(p) for p in parts)
    assert bench.rank_affinity(range(8), 3, 8) is None and bench.rank_affinity(range(15), 0, 8) is None
    assert bench.rank_affinity(range(64), 9, 8) == list(range(8, 16))          # (a global rank used as local rank wraps instead of running off the set)


def test_launch_tables_and_their_size_limit(monkeypatch):
    """GMatcher._replays: the gims_run_ops tables are the default at every size; `launch_replay_rows` > 0 restricts them to calls of at
    most that many keypoint rows (larger batches then launch one by one).  enable_timing(stepwise=True) and GIMS_NO_REPLAY / GIMS_REPLAY
    override."""
    monkeypatch.delenv("GIMS_NO_REPLAY", raising=False)
    monkeypatch.delenv("GIMS_REPLAY", raising=False)
    m = GMatcher({}).eval()
    assert m.config['launch_replay_rows'] == 0
    for part in ("encoder", "layers"):
        assert m._replays(part, 2 * 4096) and m._replays(part, 8 * 2 * 4096) and m._replays(part, 1 << 22)
    m = GMatcher({"launch_replay_rows": 16384}).eval()
    for part in ("encoder", "layers"):
        assert m._replays(part, 2 * 4096) and m._replays(part, 16384) and not m._replays(part, 16385) and not m._replays(part, 8 * 2 * 4096)
    monkeypatch.setenv("GIMS_REPLAY", "1")
    assert m._replays("layers", 65536) and m._replays("encoder", 65536)
    monkeypatch.setenv("GIMS_NO_REPLAY", "1")
    assert not m._replays("layers", 1024) and not m._replays("encoder", 1024)
    monkeypatch.setenv("GIMS_NO_REPLAY", "2")
    assert m._replays("layers", 65536) and not m._replays("encoder", 1024)
    monkeypatch.setenv("GIMS_NO_REPLAY", "3")
    assert not m._replays("layers", 1024) and m._replays("encoder", 65536)
    monkeypatch.delenv("GIMS_NO_REPLAY")
    monkeypatch.delenv("GIMS_REPLAY")
    m._stepwise = True
    assert not m._replays("layers", 1024)


def test_bench_event_sampling_schedule():
    """bench.py records its HIP events on every 5th timed step, mid-stride (bracketing every launch of every step costs 2 % of the step)."""
    import bench
    assert bench.event_steps(20, 5) == (5, [2, 7, 12, 17])
    assert bench.event_steps(5, 5) == (5, [2]) and bench.event_steps(3, 5) == (3, [1]) and bench.event_steps(1, 5) == (1, [0])
    assert bench.event_steps(20, 1) == (1, list(range(20))) and bench.event_steps(7, 0) == (1, list(range(7)))

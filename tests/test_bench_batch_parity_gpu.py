"""Parity of the TIMED path itself: the batches `bench.py` times -- `GMatcher.match_pairs` over 8 pairs of 2 x 4096 synthetic keypoints (seeds
1000 ... 1007), default 'auto' attention with its device-side guards, on-chip Sinkhorn -- against the reference's own outputs for the pairs of
that batch that have a golden (tests/golden/e2e_n4096_s1000_*_i100, e2e_n4096_s1001_*_i20: the unmodified /root/reference/models/gmatcher.py run
on the same pair, tools/gen_golden.py).  One pair through forward() is a launch of 8 (image, head) groups and takes the split-key / 4-wave
attention kernels; only a batch that fills the chip takes `attention8_bf16_kernel`, the dominant kernel of the bench line -- asserted here through
the library's launch counters (gims_attention_launch_counts), so that these tests pin THAT kernel and the guarded launches behind it.
Reference ops: gmatcher.py:35-39 (attention), 284-294 (selection).  Bars: every match index equal, scores within 1e-4."""
import numpy as np
import pytest
import torch

from gims_amd import GMatcher, hip, synth
from tests.helpers import compare_with_golden, load_golden, pair_to_data

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _bench_batch(kpts, n_pairs):
    """exactly bench.py's make_inputs: pair i of a step is synth.make_pair(kpts, 1000 + i), radius 15, percentile 2, min_size 7"""
    return [pair_to_data(synth.make_pair(kpts, 1000 + i), 15, 2, 7, device="cuda") for i in range(n_pairs)]


def _timed_like_bench(m, kpts, n_pairs):
    """bench.py's sequence: the first batch after the weights are loaded calibrates 'auto' (every layer on the split-bf16 kernels), the timed
    batches run the settled table with the guarded redo launches behind every bf16 layer.  Returns (outs, datas, launch counts) of a settled batch."""
    m.match_pairs(_bench_batch(kpts, n_pairs))
    torch.cuda.synchronize()
    rep = m.attention_report()
    assert rep["calibrated"] and rep["modes"] == ["bf16"] * 18, rep["modes"]
    hip.attention_launch_counts(reset=True)
    rescues = hip.sinkhorn_rescues()
    datas = _bench_batch(kpts, n_pairs)
    outs = m.match_pairs(datas)
    torch.cuda.synchronize()
    counts = hip.attention_launch_counts(reset=True)
    assert (m.sinkhorn_status() == 0).all() and hip.sinkhorn_rescues() == rescues
    assert m.attention_report()["redone"].sum() == 0            # diffuse softmaxes: no guard fired, what ran IS the bf16 tier
    return outs, datas, counts


@pytest.mark.parametrize("iters,thr,slot,name", [(100, 0.2, 0, "e2e_n4096_s1000_r15p2m7_i100"), (20, 0.02, 1, "e2e_n4096_s1001_r15p2m7_i20")])
def test_timed_4096x8_batch_vs_reference_golden(synth_sd, iters, thr, slot, name):
    """The headline batch (and the eval-setting block) of bench.py: 8 x 2 x 4096 through match_pairs; the pair in `slot` against the reference."""
    g = load_golden(name)
    assert [int(x) for x in g["meta"]] == [4096, 1000 + slot, 15, 2, 7, iters] and abs(float(g["match_threshold"]) - thr) < 1e-9
    m = GMatcher({"sinkhorn_iterations": iters, "match_threshold": thr}).eval()
    m.load_state_dict(synth_sd)
    outs, datas, counts = _timed_like_bench(m, 4096, 8)
    # 18 layers on the 8-wave bf16 kernel, each followed by a guarded split-bf16 attention launch; nothing on the small-launch kernels
    assert counts == dict(wave4=0, split=0, wave8=18, wave8_f16=0, x3=0, x3_guarded=18), counts
    assert m.sinkhorn_plan_last > 0                                  # the on-chip Sinkhorn (8 problems of 4096 x 4096: 4 launches)
    stats = compare_with_golden(outs[slot], datas[slot], g, thr)
    print(name, "slot", slot, stats, counts)
    assert stats["n"] == 4096


def test_timed_1024_batch_on_the_8wave_kernel_vs_reference_golden(synth_sd, monkeypatch):
    """8 x 2 x 1024 forced onto the 8-wave kernel (GIMS_ATTN_QP=8; bench.py's 32-pair batch takes it by itself) against e2e_n1024_s1000."""
    name = "e2e_n1024_s1000_r15p2m7_i100"
    g = load_golden(name)
    monkeypatch.setenv("GIMS_ATTN_QP", "8")
    m = GMatcher({}).eval()
    m.load_state_dict(synth_sd)
    outs, datas, counts = _timed_like_bench(m, 1024, 8)
    assert counts["wave8"] == 18 and counts["wave4"] == counts["split"] == 0, counts
    print(name, compare_with_golden(outs[0], datas[0], g, 0.2), counts)


def test_bench_second_workload_takes_the_8wave_kernel(synth_sd):
    """bench.py's 2 x 1024 x 32 block: its launch shape selects the 8-wave kernel without forcing; pair 0 against e2e_n1024_s1000."""
    g = load_golden("e2e_n1024_s1000_r15p2m7_i100")
    m = GMatcher({}).eval()
    m.load_state_dict(synth_sd)
    outs, datas, counts = _timed_like_bench(m, 1024, 32)
    assert counts["wave8"] == 18 and counts["wave4"] == counts["split"] == 0, counts
    print(compare_with_golden(outs[0], datas[0], g, 0.2), counts)

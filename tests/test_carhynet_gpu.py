"""GPU parity of the CAR-HyNet descriptor path: HIP kernels vs the CPU oracle and vs the reference's goldens.
Tolerance: the 3x3 / 8x8 convolutions run as split-bf16x3 GEMMs (2^-17 relative per product, seven of them in sequence with an
FRN normalisation after each), everything else in f32.  Descriptors are unit vectors (components up to ~0.4): the bar is
3e-5 absolute; measured worst case 2.3e-5 over 256 patches (tools/carhynet_bench.py), ~1e-5 typical."""
import glob
import os

import numpy as np
import pytest
import torch

from gims_amd import synth
from oracle import carhynet_oracle as CO

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "carhynet_*.npz")))


def _model(seed_w):
    from gims_amd.carhynet import CARHyNet
    m = CARHyNet().eval()
    m.load_state_dict(synth.make_carhynet_state_dict(seed_w))
    return m


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(g)[:-4] for g in GOLD])
def test_descriptors_vs_reference_golden(path):
    g = np.load(path)
    m = _model(int(g["seed_w"]))
    patches = synth.make_patches(int(g["n"]), int(g["seed_p"]))
    desc = m.compute_des_batches(patches, color=True)
    np.testing.assert_allclose(desc, g["desc"], atol=3e-5, rtol=0)
    x = torch.from_numpy(patches).permute(0, 3, 1, 2).cuda()              # the reference's NCHW forward()
    d2, raw = m(x, mode="train")
    np.testing.assert_allclose(d2.cpu().numpy(), g["desc"], atol=3e-5, rtol=0)
    np.testing.assert_allclose(raw.cpu().numpy(), g["raw"], atol=2e-4, rtol=1e-4)


def test_vs_oracle_ragged_batch():
    """A batch that is not a multiple of anything, and chunked processing (chunk smaller than the batch)."""
    m = _model(323)
    m.chunk = 37
    patches = synth.make_patches(101, 11)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_carhynet_state_dict(323).items()}
    ref, _ = CO.car_hynet_forward(sd, torch.from_numpy(patches))
    out = m.compute_des_batches(patches)
    np.testing.assert_allclose(out, ref.numpy(), atol=3e-5, rtol=0)
    np.testing.assert_allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)


def test_fused_blocks_equal_layer_by_layer():
    """gims_ch_sandglass / gims_ch_frn_block (one workgroup per patch, activation resident in LDS) against the separate kernels they replace."""
    patches = synth.make_patches(70, 31)
    outs = []
    for fused in (True, False):
        m = _model(321)
        m.fused_sandglass = m.fused_frn = fused
        m.fused_conv = False                        # same convolution GEMMs on both sides: this test isolates the two f32 blocks
        d, raw = m(torch.from_numpy(patches).permute(0, 3, 1, 2).cuda(), mode="train")
        outs.append((d.cpu().numpy(), raw.cpu().numpy()))
    np.testing.assert_allclose(outs[0][1], outs[1][1], atol=1e-4, rtol=1e-5)       # same f32 arithmetic, different summation orders (raw values up to ~3)
    np.testing.assert_allclose(outs[0][0], outs[1][0], atol=5e-6, rtol=0)


def test_fused_conv_blocks_equal_gemm_plus_frn_block():
    """gims_ch_conv_block (3x3 convolution as an implicit GEMM on the LDS-resident patch + FRN (+CoordAtt) + TLU in one kernel, layers
    2-6) against the gather-mode GEMM followed by gims_ch_frn_block: the same split-bf16x3 products, summed in a different order."""
    patches = synth.make_patches(70, 33)
    outs = []
    for fused in (True, False):
        m = _model(321)
        m.fused_conv = fused
        d, raw = m(torch.from_numpy(patches).permute(0, 3, 1, 2).cuda(), mode="train")
        outs.append((d.cpu().numpy(), raw.cpu().numpy()))
    np.testing.assert_allclose(outs[0][1], outs[1][1], atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(outs[0][0], outs[1][0], atol=1e-5, rtol=0)


def test_conv_block_layer_vs_conv2d():
    """One layer in isolation, every geometry: gims_ch_conv_block against torch's float64 conv2d + the FRN / TLU formulas on the
    SAME split-bf16 input (hi + lo) and split weights."""
    from gims_amd import hip
    r = np.random.default_rng(5)
    for hin, cin, cout, stride in ((32, 32, 32, 1), (32, 32, 64, 2), (16, 64, 64, 1), (16, 64, 128, 2), (8, 128, 128, 1)):
        n = 5
        x = r.normal(size=(n, hin, hin, cin)).astype(np.float32)
        w = (r.normal(size=(cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
        b = r.normal(size=cout).astype(np.float32) * 0.1
        fw, fb = (1 + 0.2 * r.normal(size=cout)).astype(np.float32), (0.1 * r.normal(size=cout)).astype(np.float32)
        tau = (-1 + 0.3 * r.normal(size=cout)).astype(np.float32)
        xs = hip.split_spl32(torch.from_numpy(x.reshape(-1, cin)).cuda())
        xh, xl = hip.spl32_planes(xs)
        x_seen = (xh.double() + xl.double()).cpu().reshape(n, hin, hin, cin)
        wp = hip.pack_conv3_fragments(torch.from_numpy(w))
        w_seen = (wp[:, :, 0].double() + wp[:, :, 1].double())            # [step][nb][lane][8] -> back to [cout][cin][3][3]
        w_seen = w_seen.reshape(9, cin // 16, cout // 32, 2, 32, 8).permute(2, 4, 1, 3, 5, 0).reshape(cout, cin, 3, 3)
        L = dict(wp=wp.cuda(), b=torch.from_numpy(b).cuda())
        F = dict(w=torch.from_numpy(fw).cuda(), b=torch.from_numpy(fb).cuda(), eps=1e-6)
        ho = (hin - 1) // stride + 1
        y = torch.empty((n, ho, ho, cout), dtype=torch.float32, device="cuda")
        hip.ch_conv_block(xs, n, hin, cin, cout, stride, L, F, torch.from_numpy(tau).cuda(), None, y=y)
        conv = torch.nn.functional.conv2d(x_seen.permute(0, 3, 1, 2), w_seen, torch.from_numpy(b).double(), stride=stride, padding=1)
        nu2 = (conv * conv).mean(dim=(2, 3), keepdim=True)
        ref = torch.maximum(conv * torch.rsqrt(nu2 + 1e-6) * torch.from_numpy(fw).double()[None, :, None, None]
                            + torch.from_numpy(fb).double()[None, :, None, None], torch.from_numpy(tau).double()[None, :, None, None])
        err = (y.cpu().double() - ref.permute(0, 2, 3, 1)).abs().max().item()
        assert err < 2e-5, (hin, cin, cout, stride, err)


@pytest.mark.parametrize("geom", [(32, 32, 1, True), (32, 32, 1, False), (32, 64, 2, False)])
def test_conv_block_two_per_cu_equals_one_per_cu(monkeypatch, geom):
    """The 32 x 32 layers in their two-workgroups-per-CU form (ch_conv_block_half_kernel: the patch goes through LDS in two halves, the FRN /
    CoordAtt / TLU block runs on half images) against the one-workgroup kernel (GIMS_CH_HALF=0): same products in the same order; the FRN
    statistic and the column pools are summed over the two halves in a different association, so agreement is to f32 rounding (1e-5 of values
    of order 1), with and without CoordAtt gates, f32 and split-bf16 outputs, and a patch count that leaves workgroups of the last wave idle."""
    from gims_amd import hip
    cin, cout, stride, gates = geom
    r = np.random.default_rng(17)
    n, hin = 37, 32
    ho = (hin - 1) // stride + 1
    x = r.normal(size=(n, hin, hin, cin)).astype(np.float32)
    x[3] *= 40.0                                                       # one patch of a very different scale (per-patch statistics must not mix)
    w = (r.normal(size=(cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).cuda()      # noqa: E731
    L = dict(wp=hip.pack_conv3_fragments(torch.from_numpy(w)).cuda(), b=dev(r.normal(size=cout) * 0.1))
    F = dict(w=dev(1 + 0.2 * r.normal(size=cout)), b=dev(0.1 * r.normal(size=cout)), eps=1e-6)
    tau = dev(-1 + 0.3 * r.normal(size=cout))
    G = None
    if gates:
        G = dict(w1=dev(r.normal(size=(8, cout)) / 6), b1=dev(r.normal(size=8) * 0.1), wh=dev(r.normal(size=(cout, 8)) / 3), bh=dev(r.normal(size=cout) * 0.1),
                 ww=dev(r.normal(size=(cout, 8)) / 3), bw=dev(r.normal(size=cout) * 0.1))
    xs = hip.split_spl32(torch.from_numpy(x.reshape(-1, cin)).cuda())
    outs = {}
    for half in ("0", "1"):
        monkeypatch.setenv("GIMS_CH_HALF", half)
        y = torch.full((n, ho, ho, cout), float("nan"), dtype=torch.float32, device="cuda")
        ysp = torch.zeros((n * ho * ho, 2 * cout), dtype=torch.bfloat16, device="cuda")
        hip.ch_conv_block(xs, n, hin, cin, cout, stride, L, F, tau, G, y=y)
        hip.ch_conv_block(xs, n, hin, cin, cout, stride, L, F, tau, G, y_split=ysp)
        hi, lo = hip.spl32_planes(ysp)
        outs[half] = (y.cpu().numpy(), (hi.float() + lo.float()).cpu().numpy().reshape(n, ho, ho, cout))
    scale = np.abs(outs["0"][0]).max(axis=(1, 2, 3), keepdims=True)
    assert np.isfinite(outs["1"][0]).all()
    assert (np.abs(outs["1"][0] - outs["0"][0]) / scale).max() < 1e-5
    assert (np.abs(outs["1"][1] - outs["1"][0]) <= np.abs(outs["1"][0]) * 2.0 ** -15 + 1e-30).all()      # the split output carries the same values


def test_full_size_properties():
    """BASELINE config 5's size (descriptors for 2 x 8192 keypoints), where the CPU restatement is too slow: every patch is
    processed independently, so (1) the result does not depend on how the batch is chunked -- bit for bit --, (2) a patch
    gives the same descriptor wherever it sits in the batch, (3) all descriptors are unit vectors, (4) two runs agree bitwise."""
    m = _model(321)
    base = synth.make_patches(512, 21)
    patches = np.tile(base, (32, 1, 1, 1))                      # 16384 patches, period 512
    d1 = m.compute_des_batches(patches)
    assert d1.shape == (16384, 128)
    np.testing.assert_allclose(np.linalg.norm(d1, axis=1), 1.0, atol=1e-5)
    np.testing.assert_array_equal(d1[:512], d1[512 * 17:512 * 18])
    m.chunk = 1536                                              # does not divide the batch
    d2 = m.compute_des_batches(patches)
    np.testing.assert_array_equal(d1, d2)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_carhynet_state_dict(321).items()}
    ref, _ = CO.car_hynet_forward(sd, torch.from_numpy(base[:32]))
    np.testing.assert_allclose(d1[:32], ref.numpy(), atol=3e-5, rtol=0)


def test_state_dict_roundtrip_and_errors():
    from gims_amd.carhynet import CARHyNet
    m = _model(321)
    sd = m.state_dict()
    assert list(sd.keys()) == [n for n, _ in synth.carhynet_state_dict_spec()]
    m2 = CARHyNet().eval()
    m2.load_state_dict(sd)
    p = synth.make_patches(4, 2)
    np.testing.assert_array_equal(m.compute_des_batches(p), m2.compute_des_batches(p))
    with pytest.raises(RuntimeError):
        m2.load_state_dict({"layer1.0.weight": torch.zeros(1, 3, 1, 1)})
    with pytest.raises(Exception):
        m(torch.zeros(2, 3, 32, 32))                # CPU tensor: no CPU fallback

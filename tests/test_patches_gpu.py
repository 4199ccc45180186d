"""GPU parity of the patch-extraction kernels (gims_amd/csrc/patches.hip) against oracle/patch_oracle.py:
uint8 / fixed-point work, so the bar is BIT-EXACT -- every pyramid level and every patch value.  (Both restate OpenCV's
algorithms; parity against OpenCV itself is unpinned, see the oracle's header.)  Plus the chain the reference's front end
runs after keypoint detection -- patches -> CAR-HyNet descriptors -> matcher (BASELINE config 5's "descriptor extraction fused
into the HIP path") -- against the three oracles chained."""
from collections import namedtuple

import numpy as np
import pytest
import torch

from gims_amd import Matching, frontend, hip, synth
from oracle import carhynet_oracle as CO
from oracle import gims_oracle as O
from oracle import patch_oracle as P
from tests.helpers import safe_rows

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
KP = namedtuple("KP", "pt size angle response octave")


def texture(h, w, seed):
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, 3))
    for c in range(3):
        for _ in range(12):
            fx, fy, ph, a = r.uniform(0.02, 0.6), r.uniform(0.02, 0.6), r.uniform(0, 6.28), r.uniform(5, 30)
            img[..., c] += a * np.sin(fx * xx + fy * yy + ph)
    return np.clip(128 + img + r.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)


def random_keypoints(n, h, w, seed, octaves=(-1, 0, 1), n_levels=None):
    r = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        o = int(r.choice(octaves))
        layer = int(r.integers(0, 4))
        packed = (o & 0xFF) | (layer << 8) | (int(r.integers(0, 256)) << 16)        # the xi byte is ignored by the unpacking
        out.append(KP((np.float32(r.uniform(0, w)), np.float32(r.uniform(0, h))), np.float32(r.uniform(1.5, 12.0)),
                      np.float32(r.choice([0.0, 90.0, 359.99999, r.uniform(0, 360)])), np.float32(r.uniform()), packed))
    return out


def test_keypoint_affine_equals_the_reference_bit_for_bit(golden_dir):
    """The per-keypoint arithmetic inside gims_patch_extract (one lane per keypoint: octave unpacking, the float32 / float64 mix
    of the 2x3 map, the pyramid level) through gims_patch_affine, against what the REFERENCE's ComputePatches handed to
    cv2.warpAffine for the same keypoints (tests/golden/patch_affine_*.npz, tools/gen_golden_patches.py): bit for bit."""
    import os
    names = sorted(f for f in os.listdir(golden_dir) if f.startswith("patch_affine_"))
    assert names
    for name in names:
        g = np.load(os.path.join(golden_dir, name))
        kp4 = g["kp4"].astype(np.float32)
        assert (kp4.astype(np.float64) == g["kp4"]).all()                 # cv2.KeyPoint attributes are C floats
        A, level = hip.patch_affine(torch.from_numpy(kp4).cuda(), torch.from_numpy(g["packed_octave"].astype(np.int32)).cuda())
        np.testing.assert_array_equal(level.cpu().numpy(), g["level"])
        A = A.cpu().numpy()
        bad = np.nonzero((A.view(np.int64) != g["A"].view(np.int64)).any(axis=(1, 2)))[0]
        assert len(bad) == 0, f"{len(bad)} of {len(A)} matrices differ from the reference's, e.g. #{bad[0]}: {A[bad[0]]} vs {g['A'][bad[0]]}"


def test_pyramid_layout_equals_the_reference_schedule(golden_dir):
    """gims_pyramid_layout: the number of levels and every level's size against what the reference's buildGaussianPyramid
    produced for the same image sizes (recorded call sequence, tests/golden/patch_pyramid_calls.npz)."""
    import os
    g = np.load(os.path.join(golden_dir, "patch_pyramid_calls.npz"))
    for (h, w) in g["shapes"]:
        levels, _, _ = hip.pyramid_layout(int(h), int(w), 3)
        assert len(levels) == int(g[f"{h}x{w}/n_levels"])
        assert [[L.h, L.w] for L in levels] == g[f"{h}x{w}/level_shapes"].tolist()


@pytest.mark.parametrize("h,w,seed", [(96, 128, 0), (75, 101, 1), (33, 250, 2)])
def test_pyramid_bit_exact(h, w, seed):
    img = texture(h, w, seed)
    ref = P.build_pyramid(img)
    pyr, levels, _ = hip.pyramid_build(torch.from_numpy(img).cuda())
    assert len(levels) == len(ref)
    buf = pyr.cpu().numpy()
    for i, (L, r) in enumerate(zip(levels, ref)):
        assert (L.h, L.w) == r.shape[:2], i
        got = buf[L.offset:L.offset + L.h * L.w * 3].reshape(L.h, L.w, 3)
        assert (got == r).all(), f"level {i}: {int((got != r).sum())} of {r.size} bytes differ"


@pytest.mark.parametrize("h,w,seed,n", [(120, 160, 3, 300), (64, 96, 4, 77)])
def test_patches_bit_exact(h, w, seed, n):
    img = texture(h, w, seed)
    kps = random_keypoints(n, h, w, seed + 100)
    ref = P.compute_patches([(k.pt[0], k.pt[1], k.size, k.angle, k.octave) for k in kps], P.build_pyramid(img))
    got = frontend.extract_patches(img, kps, "cuda").cpu().numpy()
    assert got.shape == ref.shape == (n, 32, 32, 3) and got.dtype == np.float32
    bad = int((got != ref).sum())
    assert bad == 0, f"{bad} of {ref.size} patch values differ (max abs diff {np.abs(got - ref).max()})"
    assert ref.max() <= 1.0 and ref.std() > 0.05            # not a degenerate comparison


def test_keypoint_outside_the_pyramid_raises_like_list_indexing():
    img = texture(40, 40, 5)
    with pytest.raises(IndexError):
        frontend.extract_patches(img, [KP((5.0, 5.0), 3.0, 0.0, 1.0, 9)], "cuda")      # octave 9 does not exist for a 40x40 image
    assert frontend.extract_patches(img, [], "cuda").shape == (0, 32, 32, 3)


def _scene(n, seed, shift=(7, 4)):
    """Two views of one texture: image 1 is image 0 translated by whole pixels, its keypoints are image 0's translated (a
    permutation of them, some dropped, some replaced), so the two descriptor sets correlate and the matcher has work to do."""
    w, h = synth.canvas_for(n)
    big = texture(h + 40, w + 40, seed)
    img0 = np.ascontiguousarray(big[20:20 + h, 20:20 + w])
    img1 = np.ascontiguousarray(big[20 - shift[1]:20 - shift[1] + h, 20 - shift[0]:20 - shift[0] + w])      # content moved by +shift
    pair = synth.make_pair(n, seed, outlier_frac=0.1)
    r = np.random.default_rng(seed)
    size = r.uniform(1.5, 3.0, n).astype(np.float32)
    ang = r.uniform(0, 360, n).astype(np.float32)
    octv = np.where(r.uniform(size=n) < 0.5, 0xFF | (1 << 8), 0xFF | (2 << 8)).astype(np.int32)      # octave -1, layers 1 / 2
    k0 = pair["keypoints0"][0]
    kps0 = [KP((k0[i, 0], k0[i, 1]), size[i], ang[i], pair["scores0"][0, i], int(octv[i])) for i in range(n)]
    gt = pair["gt_perm"]
    k1 = pair["keypoints1"][0].copy()
    size1, ang1, oct1 = r.uniform(1.5, 3.0, n).astype(np.float32), r.uniform(0, 360, n).astype(np.float32), octv.copy()
    for i in range(n):
        if gt[i] >= 0:
            j = gt[i]
            k1[j] = k0[i] + np.float32(shift)
            size1[j], ang1[j], oct1[j] = size[i], ang[i], octv[i]
    kps1 = [KP((k1[j, 0], k1[j, 1]), size1[j], ang1[j], pair["scores1"][0, j], int(oct1[j])) for j in range(n)]
    return img0, img1, kps0, kps1


def test_patches_descriptors_matcher_chain_vs_chained_oracles(synth_sd):
    """image + keypoints -> patches (f4) -> CAR-HyNet descriptors (f1) -> duplicated to 256-d (common.py:891) -> matcher:
    ``Matching`` with ``frontend.sift_forward_device`` as its front end (detection replaced by given keypoints), against
    patch_oracle -> carhynet_oracle -> gims_oracle on the same inputs."""
    from gims_amd.carhynet import CARHyNet
    n = 256
    img0, img1, kps0, kps1 = _scene(n, 21)
    net = CARHyNet().eval()
    csd = synth.make_carhynet_state_dict(321)
    net.load_state_dict(csd)
    queue = [kps0, kps1]
    m = Matching({"front_end": lambda d, device: frontend.sift_forward_device(d, device, detector=lambda img: queue.pop(0))}).eval()
    m.gmodel.load_state_dict(synth_sd)
    dev = torch.device("cuda")
    out = m({"image0": img0[None], "image1": img1[None], "carhynet": net, "device": dev, "radius": 15, "percentile": 2, "min_size": 7})
    # ---- the oracles, chained
    tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in csd.items()}
    data = {"image0": img0[None], "image1": img1[None], "device": torch.device("cpu"), "radius": 15, "percentile": 2, "min_size": 7}
    descs = {}
    for s, (img, kps) in enumerate(((img0, kps0), (img1, kps1))):
        patches = P.compute_patches([(k.pt[0], k.pt[1], k.size, k.angle, k.octave) for k in kps], P.build_pyramid(img))
        d, _ = CO.car_hynet_forward(tsd, torch.from_numpy(patches))
        descs[s] = d
        data[f"keypoints{s}"] = torch.tensor([[k.pt for k in kps]], dtype=torch.float32)
        data[f"scores{s}"] = torch.tensor([[k.response for k in kps]], dtype=torch.float32)
        data[f"descriptors{s}"] = torch.cat([d, d], 1).permute(1, 0)[None].contiguous()
    # descriptors of the HIP chain (before the adaptive graph filtered them) against the oracle chain: same bar as f1
    hip_d0 = frontend.sift_forward_device({"image": img0[None], "carhynet": net, "max_keypoints": -1}, dev, detector=lambda img: kps0)["descriptors"][0]
    np.testing.assert_allclose(hip_d0[:128].t().cpu().numpy(), descs[0].numpy(), atol=3e-5, rtol=0)
    st = {}
    ref = O.gmatcher_forward(synth_sd, data, {}, stages=st)
    assert out["keypoints0"].shape[1] == ref["keypoints0"].shape[1] and out["keypoints1"].shape[1] == ref["keypoints1"].shape[1], \
        "the adaptive graph kept different keypoints (descriptor drift moved an edge across the percentile threshold)"
    np.testing.assert_array_equal(out["keypoints0"].cpu().numpy(), ref["keypoints0"].numpy())
    r0 = ref["matches0"][0].numpy()
    safe = safe_rows(st["ot"][0].numpy(), 0.2, r0, ref["matching_scores0"][0].numpy())
    m0 = out["matches0"][0].cpu().numpy()
    assert safe.mean() > 0.9 and int((r0 >= 0).sum()) > n // 3, (safe.mean(), int((r0 >= 0).sum()))
    np.testing.assert_array_equal(m0[safe], r0[safe])
    err = np.abs(out["matching_scores0"][0].cpu().numpy() - ref["matching_scores0"][0].numpy())[m0 == r0].max()
    assert err < 2e-4, err           # descriptors enter with 3e-5 of drift (f1's bar); the matcher's own bar is 1e-4
    print("chain: matched", int((m0 >= 0).sum()), "of", len(m0), "score err", float(err))

"""The patch-extraction oracle (oracle/patch_oracle.py), CPU only.
  * PINNED against the reference: everything the reference computes itself before it calls OpenCV -- octave unpacking, the
    2x3 map / level / size / flags handed to cv2.warpAffine, octave count, call order and sigma schedule of the pyramid --
    against tests/golden/patch_affine_*.npz and patch_pyramid_calls.npz (tools/gen_golden_patches.py: the reference's own
    functions run against an argument-recording cv2 stand-in);
  * known-answer tests of the OpenCV resamplers it restates (closed-form properties; OpenCV is not available, so that part
    has no golden vector: parity unpinned, see the oracle's header)."""
import os

import numpy as np
import pytest

from oracle import patch_oracle as P

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(f for f in os.listdir(GOLD) if f.startswith("patch_affine_")))
def test_keypoint_affine_equals_the_reference_bit_for_bit(name):
    """unpackSIFTOctave (library.py:16-35) and the arguments ComputePatches gives cv2.warpAffine (library.py:96-108): 220
    keypoints over octaves -1..3, layers 0..3, angles 0 / 360 - eps / arbitrary."""
    g = np.load(os.path.join(GOLD, name))
    assert len(g["kp4"]) >= 200 and set(g["octave_layer"][:, 0]) == {-1, 0, 1, 2, 3} and set(g["octave_layer"][:, 1]) == {0, 1, 2, 3}
    assert list(g["A_dtype"]) == ["float64"] and g["dsize"].tolist() == [[64, 64]]
    assert g["flags"].tolist() == [2] and g["border"].tolist() == [0]            # INTER_CUBIC, BORDER_CONSTANT
    assert g["patch_shape"].tolist() == [64, 64, 3] and str(g["patch_dtype"]) == "float32"
    for i, (kp, packed) in enumerate(zip(g["kp4"], g["packed_octave"])):
        octave, layer, scale = P.unpack_octave(int(packed))
        assert (octave, layer) == tuple(g["octave_layer"][i]) and scale == g["scale"][i]
        A, level = P.keypoint_affine(kp[:2], kp[2], kp[3], int(packed), radius_size=64)
        assert level == g["level"][i]
        assert A.dtype == np.float64 and A.tobytes() == g["A"][i].tobytes(), (i, A, g["A"][i])
    # the 360 - eps branch of library.py:99 is exercised both ways
    near = np.abs((360.0 - g["kp4"][:, 3]) - 360.0) < 1.19209e-07
    assert near.any() and (~near).any()


def test_pyramid_schedule_equals_the_reference():
    """buildGaussianPyramid (library.py:234-271): number of levels, the order of resize / blur calls, level sizes, resize
    arguments and the sigma of every blur, as recorded from the reference (running under NumPy 2: `numpy2` promotion)."""
    g = np.load(os.path.join(GOLD, "patch_pyramid_calls.npz"))
    assert str(g["numpy_version"]).startswith("2.")
    nep = P.layer_sigmas("numpy2")
    leg = P.layer_sigmas("numpy1")
    for (h, w) in g["shapes"]:
        key = f"{h}x{w}/"
        n_oct = P.n_octaves(2 * h, 2 * w)
        assert g[key + "n_levels"] == 6 * n_oct
        # call order: one 2x INTER_LINEAR_EXACT upsampling, then per octave [resize INTER_NEAREST unless first] + 5 blurs
        kinds = [0] + [k for o in range(n_oct) for k in (([] if o == 0 else [0]) + [1] * 5)]
        assert g[key + "kinds"].tolist() == kinds
        rs = g[key + "resize"]                        # fx, fy, interpolation, src h, src w, dst h, dst w
        assert rs[0].tolist() == [2.0, 2.0, 5.0, h, w, 2 * h, 2 * w]
        assert (rs[1:, :3] == [0.5, 0.5, 0.0]).all()
        # level sizes, incl. cvRound(n / 2) of odd sizes: the oracle's own pyramid for the small images, its size rule for the rest
        if h * w <= 320 * 240:
            shapes = [list(p.shape[:2]) for p in P.build_pyramid(np.zeros((h, w, 3), dtype=np.uint8))]
        else:
            shapes, hh, ww = [], 2 * h, 2 * w
            for o in range(n_oct):
                if o:
                    hh, ww = P.half_nearest(np.zeros((hh, ww, 1), np.uint8)).shape[:2]
                shapes += [[hh, ww]] * 6
        assert shapes == g[key + "level_shapes"].tolist()
        assert g[key + "blur_ksize"].tolist() == [[0, 0]] and list(g[key + "blur_sigma_type"]) == ["float32"]
        sig = g[key + "blur_sigma"]
        assert (sig == g[key + "blur_sigma_y"]).all()
        want = np.array(nep[1:] * n_oct, dtype=np.float64)
        assert sig.tobytes() == want.tobytes()                                           # bit for bit
    # the product uses the reference's PINNED numpy (1.26: float64 products, library.py:252-257) -- same kernels either way
    for a, b in zip(leg[1:], nep[1:]):
        assert abs(a - b) <= 3e-7 * a
        assert (P.gaussian_kernel_q8(a) == P.gaussian_kernel_q8(b)).all()


def _img(h, w, seed=0):
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 120 + 60 * np.sin(xx / 7.0) * np.cos(yy / 5.0)
    return np.clip(base[..., None] + r.normal(0, 20, (h, w, 3)), 0, 255).astype(np.uint8)


def test_gaussian_kernels_are_normalised_and_symmetric():
    sig = P.layer_sigmas()
    assert abs(sig[1] - 1.2262734984654078) < 1e-6 and abs(sig[5] - 3.0900155872894388) < 1e-6      # sqrt((1.6 k^i)^2 - (1.6 k^(i-1))^2)
    for s, n in zip(sig[1:], (9, 11, 13, 17, 21)):        # cvRound(sigma * 6 + 1) | 1
        q = P.gaussian_kernel_q8(s)
        assert len(q) == n and q.sum() == 256 and (q == q[::-1]).all() and q.argmax() == n // 2 and (q >= 0).all()


def test_blur_and_resize_preserve_constants_and_shapes():
    c = np.full((37, 53, 3), 77, dtype=np.uint8)
    assert (P.gaussian_blur_u8(c, 1.6) == 77).all()
    assert (P.up2x_linear_exact(c) == 77).all() and P.up2x_linear_exact(c).shape == (74, 106, 3)
    assert P.half_nearest(np.zeros((75, 101, 3), np.uint8)).shape == (38, 50, 3)          # cvRound: half to even
    img = _img(24, 32)
    up = P.up2x_linear_exact(img).astype(int)
    a = img.astype(int)
    # interior destination pixel (2k+1, 2l+1) = (9 a[k,l] + 3 a[k,l+1] + 3 a[k+1,l] + a[k+1,l+1] + 8) >> 4
    k, l = 5, 9
    assert (up[2 * k + 1, 2 * l + 1] == (9 * a[k, l] + 3 * a[k, l + 1] + 3 * a[k + 1, l] + a[k + 1, l + 1] + 8) >> 4).all()
    assert (P.half_nearest(img) == img[::2, ::2]).all()


def test_pyramid_structure():
    img = _img(60, 80, 1)
    pyr = P.build_pyramid(img)
    assert len(pyr) % 6 == 0 and len(pyr) // 6 == int(np.round(np.log2(120) - 2)) + 1        # library.py:248
    assert pyr[0].shape == (120, 160, 3) and pyr[6].shape == (60, 80, 3) and pyr[12].shape == (30, 40, 3)
    assert (pyr[6] == pyr[3][::2, ::2]).all()                                                 # library.py:264-265
    # blurring lowers the high-frequency energy monotonically inside an octave
    e = [np.abs(np.diff(p.astype(int), axis=1)).mean() for p in pyr[:6]]
    assert all(e[i + 1] < e[i] for i in range(5))


def test_cubic_table_and_identity_warp():
    tab = P.cubic_weight_table()
    assert tab.shape == (1024, 16) and (tab.sum(1) == 32768).all()
    # zero fraction: the pixel itself -- 1.0 * 2^15 saturates to 32767 in int16 and the missing unit lands on tap (2, 2)
    assert tab[0].tolist() == [0, 0, 0, 0, 0, 32767, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0]
    img = _img(90, 100, 2)
    # a pure integer translation reproduces the source window exactly; outside the image BORDER_CONSTANT gives 0
    M = np.array([[1.0, 0.0, -10.0], [0.0, 1.0, -20.0]])       # dst = src - (10, 20)
    out = P.warp_affine_cubic_u8(img, M, 64)
    assert (out == img[20:84, 10:74]).all()
    M2 = np.array([[1.0, 0.0, 30.0], [0.0, 1.0, 0.0]])
    out2 = P.warp_affine_cubic_u8(img, M2, 64)
    assert (out2[:, :28] == 0).all() and (out2[:, 30:] == img[:64, :34]).all()


def test_keypoint_affine_follows_the_reference_formula():
    # octave 0, layer 1 (packed 0x100), angle 0 -> 360 - 0 = 360 -> reset to 0 (library.py:99-100): pure scaling about the point
    M, level = P.keypoint_affine((50.0, 40.0), 8.0, 0.0, 0x100, 64)
    assert level == 1 * 6 + 1
    step = 8.0 * 1.0 * 0.5
    np.testing.assert_allclose(M, [[1 / step, 0, 31.5 - 50.0 / step], [0, 1 / step, 31.5 - 40.0 / step]], atol=1e-6)
    # octave -1 (packed 0xFF): scale 2, the doubled base image
    M, level = P.keypoint_affine((50.0, 40.0), 4.0, 90.0, 0xFF | (2 << 8), 64)
    assert level == 2 and abs(M[0, 1] - np.float32(-np.sin(np.deg2rad(270.0))) / np.float32(4.0)) < 1e-7
    img = _img(64, 96, 3)
    patches = P.compute_patches([(30.0, 20.0, 6.0, 33.0, 0xFF | (1 << 8)), (10.5, 50.25, 9.0, 270.0, 0x200)], P.build_pyramid(img))
    assert patches.shape == (2, 32, 32, 3) and patches.dtype == np.float32 and 0 <= patches.min() and patches.max() <= 1

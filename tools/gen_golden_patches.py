#!/usr/bin/env python3
"""Golden vectors for the part of the patch-extraction front end that the reference itself holds,
BY RUNNING THE REFERENCE ITSELF.  Build container only (needs /root/reference).

Everything that touches pixels in utils/library.py is an OpenCV call, and OpenCV is neither under /root/reference nor
installed here -- the resampling arithmetic stays unpinned.  What the reference DOES compute itself, in NumPy, is what it
hands to OpenCV:
  * unpackSIFTOctave (library.py:16-35): packed cv2.KeyPoint.octave -> (octave, layer, scale);
  * ComputePatches (library.py:84-110): per keypoint the 2x3 map ``A`` (a float32 / float64 mix), the pyramid level it warps,
    the output size and the interpolation / border flags it passes to cv2.warpAffine;
  * buildGaussianPyramid (library.py:234-271): the number of octaves, the sigma of every cv2.GaussianBlur call and the order of
    the cv2.resize / cv2.GaussianBlur calls.
The functions run UNMODIFIED against a ``cv2`` stand-in that RECORDS its arguments (and returns zero images of the size OpenCV
documents); only the recorded numbers are written -- tests/golden/patch_affine_*.npz, tests/golden/patch_pyramid_calls.npz.

NumPy note: the reference pins numpy==1.26.4 (requirements:2); this container runs NumPy 2 (NEP 50 scalar promotion), under
which library.py:252-257 evaluates the sigma schedule in float32 where 1.26 promotes to float64.  The fixture records what
runs HERE (``numpy_version`` is stored); oracle/patch_oracle.py restates both promotion rules, the test pins the NEP 50 one
bit for bit and shows that the Q8.8 blur kernels -- the only way sigma enters the arithmetic -- are the same under both.

    python tools/gen_golden_patches.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")

# ------------------------------------------------------------------------------------------------ recording cv2 stand-in
CALLS = []
cv2 = types.ModuleType("cv2")
# public enum values of OpenCV 4.x (imgproc.hpp / core/base.hpp)
cv2.INTER_NEAREST, cv2.INTER_LINEAR, cv2.INTER_CUBIC, cv2.INTER_AREA, cv2.INTER_LINEAR_EXACT = 0, 1, 2, 3, 5
cv2.BORDER_CONSTANT = 0
cv2.COLOR_BGR2GRAY = 6


def _cv_round(x):                       # cvRound: round half to even
    return int(np.rint(x))


def _resize(src, dsize, fx=0, fy=0, interpolation=1):
    h, w = src.shape[:2]
    if tuple(dsize) == (0, 0):
        nh, nw = _cv_round(h * fy), _cv_round(w * fx)        # resize.cpp: dsize = Size(saturate_cast<int>(w * fx), saturate_cast<int>(h * fy))
    else:
        nw, nh = dsize
    CALLS.append(("resize", int(getattr(src, "level", -1)), float(fx), float(fy), int(interpolation), h, w, nh, nw))
    return np.zeros((nh, nw) + src.shape[2:], dtype=src.dtype).view(Tagged)


def _blur(src, ksize, sigmaX=0, sigmaY=0):
    CALLS.append(("blur", int(getattr(src, "level", -1)), tuple(ksize), np.float64(sigmaX), np.float64(sigmaY), type(sigmaX).__name__))
    return np.zeros_like(src).view(Tagged)


def _warp(img, A, dsize, flags=1, borderMode=0):
    CALLS.append(("warp", int(img.level), np.array(A), A.dtype.name, tuple(int(x) for x in dsize), int(flags), int(borderMode)))
    return np.zeros((int(dsize[1]), int(dsize[0])) + img.shape[2:], dtype=img.dtype)


class Tagged(np.ndarray):
    """An image that remembers which pyramid level it is (set by the caller below)."""
    level = -1


cv2.resize, cv2.GaussianBlur, cv2.warpAffine = _resize, _blur, _warp
cv2.cvtColor = lambda src, code: src[..., 0]
sys.modules["cv2"] = cv2

spec = importlib.util.spec_from_file_location("ref_library", "/root/reference/utils/library.py")
L = importlib.util.module_from_spec(spec)
spec.loader.exec_module(L)                      # the reference's own file, executed


class KP:
    """The four cv2.KeyPoint attributes ComputePatches reads."""

    def __init__(self, pt, size, angle, octave):
        self.pt, self.size, self.angle, self.octave = pt, size, angle, octave


def keypoints(seed):
    r = np.random.default_rng(seed)
    kps = []
    # every (octave, layer) combination SIFT produces with nOctaveLayers = 3 (layers 1..3) plus layer 0, octaves -1..3
    for octave in (-1, 0, 1, 2, 3):
        for layer in (0, 1, 2, 3):
            # cv2.KeyPoint holds C floats: every attribute is float32-representable.  0 and 5e-8 take the `angle -> 0` branch of
            # library.py:99 (|360 - a - 360| < FLT_EPSILON), 2e-7 just misses it, nextafter(360, 0) is the largest angle below 360
            for angle in (0.0, float(np.float32(5e-8)), float(np.float32(2e-7)), float(np.nextafter(np.float32(360), np.float32(0))),
                          float(np.float32(r.uniform(0, 360))), 90.0, 180.0, float(np.float32(r.uniform(0, 360))), float(np.float32(359.99)), 45.0,
                          float(np.float32(r.uniform(0, 360)))):
                pt = (float(np.float32(r.uniform(2, 890))), float(np.float32(r.uniform(2, 660))))
                size = float(np.float32(r.uniform(1.8, 60.0)))
                xi = float(r.uniform(-0.5, 0.5))
                kps.append(KP(pt, size, angle, L.packSIFTOctave(octave, layer, xi)))
    return kps


def main():
    os.makedirs(OUT, exist_ok=True)
    # ---- ComputePatches: 220 keypoints, the reference's default call (radius_size=64, utils/common.py:883)
    for seed in (41,):
        kps = keypoints(seed)
        n_levels = 8 * 6
        gpyr = []
        for i in range(n_levels):
            im = np.zeros((4, 4, 3), dtype=np.uint8).view(Tagged)
            im.level = i
            gpyr.append(im)
        CALLS.clear()
        patches = L.ComputePatches(kps, gpyr, radius_size=64)
        assert len(CALLS) == len(kps) and all(c[0] == "warp" for c in CALLS)
        unp = np.array([L.unpackSIFTOctave(k)[:2] for k in kps], dtype=np.int64)
        scale = np.array([L.unpackSIFTOctave(k)[2] for k in kps], dtype=np.float64)
        np.savez_compressed(
            os.path.join(OUT, f"patch_affine_s{seed}.npz"),
            kp4=np.array([[k.pt[0], k.pt[1], k.size, k.angle] for k in kps], dtype=np.float64),
            packed_octave=np.array([k.octave for k in kps], dtype=np.int64),
            octave_layer=unp, scale=scale,
            level=np.array([c[1] for c in CALLS], dtype=np.int64),
            A=np.stack([c[2] for c in CALLS]).astype(np.float64), A_dtype=np.array(sorted({c[3] for c in CALLS})),
            dsize=np.array(sorted({c[4] for c in CALLS}), dtype=np.int64), flags=np.array(sorted({c[5] for c in CALLS}), dtype=np.int64),
            border=np.array(sorted({c[6] for c in CALLS}), dtype=np.int64),
            patch_shape=np.array(patches[0].shape, dtype=np.int64), patch_dtype=np.array(patches[0].dtype.name),
            numpy_version=np.array(np.__version__))
        print(f"patch_affine_s{seed}: {len(kps)} keypoints, octaves {sorted(set(unp[:, 0]))}, layers {sorted(set(unp[:, 1]))}, "
              f"A dtype {sorted({c[3] for c in CALLS})}, dsize {sorted({c[4] for c in CALLS})}")
    # ---- buildGaussianPyramid: call sequence and sigmas for several image sizes
    rec = {}
    shapes = [(480, 640), (600, 800), (672, 896), (75, 101), (240, 320), (1080, 1920)]
    for (h, w) in shapes:
        CALLS.clear()
        base = np.zeros((h, w, 3), dtype=np.uint8).view(Tagged)
        pyr = L.buildGaussianPyramid(base, 6, graydesc=False)
        # replay with level tags: the stand-in cannot know the index an image will get, so derive the source level of call j
        # from the reference's own indexing (library.py:262-268) and check it against the shapes that were recorded
        kinds = np.array([0 if c[0] == "resize" else 1 for c in CALLS], dtype=np.int64)          # 0 resize, 1 blur
        rec[f"{h}x{w}/n_levels"] = np.int64(len(pyr))
        rec[f"{h}x{w}/kinds"] = kinds
        rec[f"{h}x{w}/level_shapes"] = np.array([p.shape[:2] for p in pyr], dtype=np.int64)
        rec[f"{h}x{w}/resize"] = np.array([(c[2], c[3], c[4], c[5], c[6], c[7], c[8]) for c in CALLS if c[0] == "resize"], dtype=np.float64)
        rec[f"{h}x{w}/blur_sigma"] = np.array([c[3] for c in CALLS if c[0] == "blur"], dtype=np.float64)
        rec[f"{h}x{w}/blur_sigma_y"] = np.array([c[4] for c in CALLS if c[0] == "blur"], dtype=np.float64)
        rec[f"{h}x{w}/blur_sigma_type"] = np.array(sorted({c[5] for c in CALLS if c[0] == "blur"}))
        rec[f"{h}x{w}/blur_ksize"] = np.array(sorted({c[2] for c in CALLS if c[0] == "blur"}), dtype=np.int64)
        print(f"pyramid {h}x{w}: {len(pyr)} levels, {int((kinds == 0).sum())} resize + {int((kinds == 1).sum())} blur calls, "
              f"sigma types {rec[f'{h}x{w}/blur_sigma_type']}")
    rec["shapes"] = np.array(shapes, dtype=np.int64)
    rec["numpy_version"] = np.array(np.__version__)
    np.savez_compressed(os.path.join(OUT, "patch_pyramid_calls.npz"), **rec)


if __name__ == "__main__":
    main()

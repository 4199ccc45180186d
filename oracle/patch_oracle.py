"""CPU ORACLE for the patch-extraction front end -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates in NumPy what the reference does between keypoint detection and the CAR-HyNet descriptor network
(utils/common.py:882-884):

    pyramid = buildGaussianPyramid(img, 6, graydesc=False)            utils/library.py:234-271
    pts     = ComputePatches(k, pyramid, radius_size=64)              utils/library.py:84-110
    pts     = [cv2.resize(p, (32, 32), interpolation=cv2.INTER_AREA) for p in pts] / 255.0

**PINNED UP TO OPENCV'S RESAMPLERS.**  What the reference computes ITSELF before it calls OpenCV -- the octave unpacking, the
2x3 map / pyramid level / size / flags ComputePatches hands to cv2.warpAffine, the octave count, call order and sigma schedule
of buildGaussianPyramid -- is pinned bit for bit by tests/golden/patch_affine_*.npz and patch_pyramid_calls.npz, recorded from
the reference's own functions running against an argument-recording cv2 stand-in (tools/gen_golden_patches.py;
tests/test_patch_oracle_cpu.py).  The pixel arithmetic itself stays **PARITY UNPINNED**: every step of it lives in OpenCV (opencv-python==4.9.0.80, requirements:3), which is neither
under /root/reference nor installed in this image: `cv2.resize(INTER_LINEAR_EXACT / INTER_NEAREST / INTER_AREA)`,
`cv2.GaussianBlur` on uint8 (fixed-point kernel with error diffusion, Q8.8 row pass, Q16.16 column pass) and
`cv2.warpAffine(INTER_CUBIC, BORDER_CONSTANT)` (inverse map in double, 1/32-pixel fixed-point coordinates, 32x32 table of
15-bit bicubic weights, A = -0.75).  The functions below restate OpenCV 4.x's published algorithm for each of them
(imgproc/src/resize.cpp, smooth.dispatch.cpp + fixedpoint.inl.hpp, imgwarp.cpp) as exactly as it can be written down without
the library at hand; the reference repository holds no test vector for them.  What IS pinned: the HIP kernels
(gims_amd/csrc/patches.hip) are bit-identical to this file on seeded inputs (tests/test_patches_gpu.py), and the host-side
keypoint arithmetic (octave unpacking, the float32 / float64 mix of the affine map) follows utils/library.py:16-35, 96-108
line by line.

Only tests/ may import this file.
"""
from __future__ import annotations

import numpy as np

N_OCTAVE_LAYERS = 3          # library.py:238
SIGMA = 1.6                  # library.py:239
FIRST_OCTAVE = -1            # library.py:240


# ------------------------------------------------------------------------------------------------ pyramid
def up2x_linear_exact(img: np.ndarray) -> np.ndarray:
    """cv2.resize(base, (0,0), fx=2, fy=2, INTER_LINEAR_EXACT) on uint8 HxWxC (library.py:245).  Source coordinate of
    destination pixel d is (d + 0.5) / 2 - 0.5: even d = 2k -> k - 0.25 (pixels k-1, k with weights 1/4, 3/4), odd d = 2k+1
    -> k + 0.25 (k, k+1 with 3/4, 1/4), indices clamped at the borders; the bit-exact path keeps all fractions (weights are
    multiples of 1/4 in each direction) and rounds once, half up: (sum of 16ths + 8) >> 4."""
    h, w = img.shape[:2]
    a = img.astype(np.int32)

    def taps(n):
        d = np.arange(2 * n)
        k = d // 2
        i0 = np.where(d % 2 == 0, k - 1, k)
        w1 = np.where(d % 2 == 0, 3, 1)            # weight of i0 + 1 in quarters
        return np.clip(i0, 0, n - 1), np.clip(i0 + 1, 0, n - 1), 4 - w1, w1

    y0, y1, wy0, wy1 = taps(h)
    x0, x1, wx0, wx1 = taps(w)
    rows = a[y0] * wy0[:, None, None] + a[y1] * wy1[:, None, None]                  # [2h][w][c], in quarters
    out = rows[:, x0] * wx0[None, :, None] + rows[:, x1] * wx1[None, :, None]        # sixteenths
    return ((out + 8) >> 4).astype(np.uint8)


def gaussian_kernel_q8(sigma: float) -> np.ndarray:
    """The Q8.8 kernel cv2.GaussianBlur uses on uint8 images with ksize = (0, 0): ksize = cvRound(sigma*3*2 + 1) | 1,
    values exp(-x^2 / (2 sigma^2)) normalised in double, then quantised to 1/256 with error diffusion from the ends
    towards the centre (cvRound = round half to even) -- the centre tap takes what is left of 256
    (getGaussianKernelFixedPoint_ED, smooth.dispatch.cpp)."""
    n = int(np.rint(sigma * 6 + 1)) | 1
    x = np.arange(n, dtype=np.float64) - (n - 1) * 0.5
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    k /= k.sum()
    q = np.zeros(n, dtype=np.int64)
    err = 0.0
    for i in range(n // 2):
        adj = k[i] * 256.0 + err
        v = int(np.rint(adj))
        err = adj - v
        q[i] = q[n - 1 - i] = v
    q[n // 2] = 256 - 2 * int(q[: n // 2].sum())
    return q


def _reflect101(i: np.ndarray, n: int) -> np.ndarray:
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.mod(i, p)
    return np.where(i >= n, p - i, i)


def gaussian_blur_u8(img: np.ndarray, sigma: float) -> np.ndarray:
    """cv2.GaussianBlur(src, (0, 0), sigmaX=sigma, sigmaY=sigma) on uint8 (library.py:268): separable, BORDER_REFLECT_101,
    row pass in Q8.8 (sum of q * pixel), column pass in Q16.16 (sum of q * row value), one rounding: (v + 2^15) >> 16."""
    q = gaussian_kernel_q8(sigma)
    r = len(q) // 2
    h, w = img.shape[:2]
    a = img.astype(np.int64)
    xs = _reflect101(np.arange(-r, w + r), w)
    rowp = np.zeros_like(a)
    for t in range(len(q)):
        rowp += q[t] * a[:, xs[t:t + w]]
    ys = _reflect101(np.arange(-r, h + r), h)
    out = np.zeros_like(a)
    for t in range(len(q)):
        out += q[t] * rowp[ys[t:t + h]]
    return np.clip((out + (1 << 15)) >> 16, 0, 255).astype(np.uint8)


def half_nearest(img: np.ndarray) -> np.ndarray:
    """cv2.resize(src, (0,0), fx=0.5, fy=0.5, INTER_NEAREST) (library.py:265): size cvRound(n * 0.5) (half to even),
    destination pixel d reads source floor(d * 2) (clamped)."""
    h, w = img.shape[:2]
    nh, nw = int(np.rint(h * 0.5)), int(np.rint(w * 0.5))
    ys = np.minimum(np.arange(nh) * 2, h - 1)
    xs = np.minimum(np.arange(nw) * 2, w - 1)
    return np.ascontiguousarray(img[ys][:, xs])


def layer_sigmas(promotion: str = "numpy1"):
    """library.py:252-257, with the scalar promotions written out so the result does not depend on the NumPy running this file.
    ``"numpy1"`` (default; what the product uses): the reference's pinned numpy==1.26.4 (requirements:2) -- 1.0 / np.float32(3)
    and pow(2.0, .) are float64 there, k is rounded to float32, pow(k, np.float32(i-1)) is a float32 power, and its product with
    the python float sigma is float64.  ``"numpy2"``: NumPy >= 2 (NEP 50: python scalars are weak) -- 1.0 / np.float32(3), the
    power, both products and the square root all stay float32.  The second is what the golden fixture records (the reference
    runs under NumPy 2 in the build container); the two schedules differ by <= 3e-7 relative and give the same Q8.8 kernels."""
    sig = [SIGMA]
    if promotion == "numpy1":
        k = np.float32(pow(2.0, 1.0 / float(np.float32(N_OCTAVE_LAYERS))))
        for i in range(1, N_OCTAVE_LAYERS + 3):
            sig_prev = float(pow(k, np.float32(i - 1))) * SIGMA            # scalar float32 power (libm powf), like the reference's pow()
            sig_total = sig_prev * float(k)
            sig.append(float(np.sqrt(sig_total * sig_total - sig_prev * sig_prev)))
        return sig
    if promotion != "numpy2":
        raise ValueError(promotion)
    f32 = np.float32
    k = f32(pow(f32(2.0), f32(1.0) / f32(N_OCTAVE_LAYERS)))
    for i in range(1, N_OCTAVE_LAYERS + 3):
        sig_prev = f32(pow(k, f32(i - 1)) * f32(SIGMA))
        sig_total = f32(sig_prev * k)
        sig.append(float(np.sqrt(f32(f32(sig_total * sig_total) - f32(sig_prev * sig_prev)))))
    return sig


def n_octaves(rows: int, cols: int) -> int:
    """library.py:248-250 on the DOUBLED image size (rows, cols): round(log(float32(min)) / log(2.0) - 2) - firstOctave."""
    return int(np.int32(np.round(np.log(np.float32(min(cols, rows))) / np.log(2.0) - 2) - FIRST_OCTAVE))


def build_pyramid(base: np.ndarray):
    """buildGaussianPyramid(base, LastOctave, graydesc=False) (library.py:234-271) on a uint8 HxWx3 image: list of
    nOctaves * 6 uint8 images.  Note: unlike OpenCV's SIFT the first image of octave 0 is the UNBLURRED doubled input."""
    base = up2x_linear_exact(base)
    rows, cols = base.shape[:2]
    n_oct = n_octaves(rows, cols)
    sig = layer_sigmas()
    L = N_OCTAVE_LAYERS + 3
    pyr = []
    for o in range(n_oct):
        for i in range(L):
            if o == 0 and i == 0:
                img = base
            elif i == 0:
                img = half_nearest(pyr[(o - 1) * L + N_OCTAVE_LAYERS])
            else:
                img = gaussian_blur_u8(pyr[o * L + i - 1], sig[i])
            pyr.append(img)
    return pyr


# ------------------------------------------------------------------------------------------------ keypoints
def unpack_octave(packed: int):
    """unpackSIFTOctave (library.py:16-35)."""
    octave = packed & 0xFF
    layer = (packed >> 8) & 0xFF
    if octave >= 128:
        octave |= -128
    scale = float(1 / (1 << octave)) if octave >= 0 else float(1 << -octave)
    return octave, layer, scale


def keypoint_affine(pt, size: float, angle: float, packed_octave: int, radius_size: int = 64):
    """The 2x3 map ComputePatches hands to cv2.warpAffine and the pyramid index it warps (library.py:96-108)."""
    flt_epsilon = 1.19209e-07
    r = (radius_size - 1) / 2
    octave, layer, scale = unpack_octave(int(packed_octave))
    step = float(size) * scale * 0.5
    ptf = np.array([float(pt[0]), float(pt[1])]) * scale
    ang = 360.0 - float(angle)
    ang = np.where(np.abs(ang - 360.0) < flt_epsilon, 0.0, ang)
    level = (octave - FIRST_OCTAVE) * (N_OCTAVE_LAYERS + 3) + layer
    phi = np.deg2rad(ang)
    s, c = np.sin(phi), np.cos(phi)
    A = np.float32([[c, -s], [s, c]]) / step
    Rptf = np.matmul(A, ptf)
    A = np.hstack([A, [[r - Rptf[0]], [r - Rptf[1]]]])
    return np.asarray(A, dtype=np.float64), int(level)


# ------------------------------------------------------------------------------------------------ warpAffine (cubic, uint8)
INTER_BITS, AB_BITS = 5, 10
INTER_TAB = 1 << INTER_BITS


def _cubic_coeffs(x: np.float32):
    A = np.float32(-0.75)
    one = np.float32(1)
    c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    return np.array([c0, c1, c2, c3], dtype=np.float32)


def cubic_weight_table() -> np.ndarray:
    """initInterTab2D(INTER_CUBIC, fixpt=true) (imgwarp.cpp): [32*32][16] int16 weights, scale 2^15, each 4x4 set forced to
    sum to 2^15 by moving the difference onto the largest (or smallest) of the four central taps."""
    tab1 = np.stack([_cubic_coeffs(np.float32(i) * np.float32(1.0 / INTER_TAB)) for i in range(INTER_TAB)])
    out = np.zeros((INTER_TAB * INTER_TAB, 16), dtype=np.int32)
    for i in range(INTER_TAB):
        for j in range(INTER_TAB):
            w = np.zeros((4, 4), dtype=np.int32)
            for k1 in range(4):
                vy = tab1[i, k1]
                for k2 in range(4):
                    v = np.float32(vy * tab1[j, k2])
                    w[k1, k2] = int(np.clip(np.rint(np.float32(v * np.float32(32768.0))), -32768, 32767))
            isum = int(w.sum())
            if isum != 32768:
                diff = isum - 32768
                Mk, mk = (2, 2), (2, 2)
                for k1 in (2, 3):
                    for k2 in (2, 3):
                        if w[k1, k2] < w[mk]:
                            mk = (k1, k2)
                        elif w[k1, k2] > w[Mk]:
                            Mk = (k1, k2)
                if diff < 0:
                    w[Mk] -= diff
                else:
                    w[mk] -= diff
            out[i * INTER_TAB + j] = w.reshape(16)
    return out


_WTAB = None


def invert_affine(M: np.ndarray) -> np.ndarray:
    """The in-place inversion at the top of cv::warpAffine (imgwarp.cpp), double precision, this operation order."""
    m = [float(v) for v in np.asarray(M, dtype=np.float64).reshape(6)]
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return np.asarray(m, dtype=np.float64)


def warp_affine_cubic_u8(img: np.ndarray, M: np.ndarray, dim: int = 64) -> np.ndarray:
    """cv2.warpAffine(img, M, (dim, dim), flags=INTER_CUBIC, borderMode=BORDER_CONSTANT) on uint8 HxWxC (library.py:107)."""
    global _WTAB
    if _WTAB is None:
        _WTAB = cubic_weight_table()
    m = invert_affine(M)
    h, w = img.shape[:2]
    AB = 1 << AB_BITS
    xs = np.arange(dim, dtype=np.float64)
    adelta = np.rint(m[0] * xs * AB).astype(np.int64)
    bdelta = np.rint(m[3] * xs * AB).astype(np.int64)
    rd = AB // INTER_TAB // 2
    ys = np.arange(dim, dtype=np.float64)
    X0 = np.rint((m[1] * ys + m[2]) * AB).astype(np.int64) + rd
    Y0 = np.rint((m[4] * ys + m[5]) * AB).astype(np.int64) + rd
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = (X >> INTER_BITS) - 1
    sy = (Y >> INTER_BITS) - 1
    wt = _WTAB[(Y & (INTER_TAB - 1)) * INTER_TAB + (X & (INTER_TAB - 1))]          # [dim][dim][16]
    a = img.astype(np.int64)
    acc = np.zeros((dim, dim, img.shape[2]), dtype=np.int64)
    for i in range(4):
        yy = sy + i
        oky = (yy >= 0) & (yy < h)
        for j in range(4):
            xx = sx + j
            ok = oky & (xx >= 0) & (xx < w)
            px = a[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]                    # [dim][dim][c]
            acc += np.where(ok[..., None], px, 0) * wt[..., 4 * i + j][..., None]
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def area_half_f32(patch_u8: np.ndarray) -> np.ndarray:
    """cv2.resize(p.astype(float32), (32, 32), INTER_AREA) for a 64x64 patch: the 2x2 mean in float32, then / 255.0
    (common.py:884)."""
    p = patch_u8.astype(np.float32)
    s = (p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2]) * np.float32(0.25)
    return (s / np.float32(255.0)).astype(np.float32)


def compute_patches(kpts, pyr, radius_size: int = 64) -> np.ndarray:
    """ComputePatches + the INTER_AREA resize + / 255 (library.py:84-110, common.py:883-884): [n][32][32][3] float32.
    kpts: iterable of (x, y, size, angle, packed_octave)."""
    dim = int(np.int32(2 * ((radius_size - 1) / 2) + 1))
    out = []
    for x, y, size, angle, octv in kpts:
        M, level = keypoint_affine((x, y), size, angle, int(octv), radius_size)
        out.append(area_half_f32(warp_affine_cubic_u8(pyr[level], M, dim)))
    return np.stack(out) if out else np.zeros((0, dim // 2, dim // 2, 3), dtype=np.float32)

"""Portable synthetic inputs and weights for the GIMS matcher hot path.

Everything here is produced by a counter-based integer hash evaluated with NumPy
uint64 arithmetic, followed only by exactly-representable float operations
(24-bit uniforms, Irwin-Hall sums), so the same seed yields the *same bits* on
the build container and on the GPU box.  No libm call (log/sin/cos) is involved
before the final L2 normalisation of descriptors, which is a plain float32
sqrt/divide (IEEE-exact).

The recipes follow SURVEY.md section 8(d) / BASELINE.md section 3:

* keypoints ~ U(canvas) with density-matched canvases so the adaptive graph
  (r=15, p=2, min_size=7) keeps (almost) every keypoint;
* descriptors: L2-normalised 128-d, duplicated to 256-d -- mirrors the
  reference's ``torch.cat([d, d])`` in utils/common.py:891;
* image 1 = permutation of image 0 + position noise + descriptor noise;
* weights: N(0, g^2 / fan_in) with per-module gains chosen so that the match
  matrix is non-degenerate (default PyTorch init collapses to zero matches).
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)

# canvas (W, H) per keypoint count -- mean radius-15 degree ~ 9
CANVAS = {64: (80, 60), 128: (112, 84), 256: (160, 120), 512: (224, 168),
          1024: (320, 240), 2048: (448, 336), 4096: (640, 480), 8192: (896, 672)}


def canvas_for(n: int):
    if n in CANVAS:
        return CANVAS[n]
    # keep density of 1024 @ 320x240
    s = (n / 1024.0) ** 0.5
    return int(round(320 * s)), int(round(240 * s))


def _mix(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        x = x ^ (x >> np.uint64(31))
    return x


def _stream_key(seed: int, stream: int) -> np.uint64:
    k = _mix(np.array([seed & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))
    k = _mix(k ^ np.uint64(stream & 0xFFFFFFFFFFFFFFFF))
    return k[0]


def uniform(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n float64 values k/2^24, k in [0, 2^24) -- exactly representable in f32."""
    key = _stream_key(seed, stream)
    with np.errstate(over="ignore"):
        ctr = (np.arange(offset, offset + n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)) & _M64
    h = _mix(ctr ^ key)
    return (h >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


def normal(seed: int, stream: int, n: int) -> np.ndarray:
    """Approximately N(0,1): Irwin-Hall sum of 4 uniforms, exact in float64."""
    u = uniform(seed, stream, 4 * n).reshape(4, n)
    s = (u[0] + u[1]) + (u[2] + u[3])          # exact: multiples of 2^-24 below 4
    return (s - 2.0) * 1.7320508075688772       # var of sum = 4/12 -> scale sqrt(3)


def permutation(seed: int, stream: int, n: int) -> np.ndarray:
    """Deterministic permutation: argsort of hash keys (stable, ties broken by index)."""
    key = _stream_key(seed, stream)
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)) & _M64
    h = _mix(ctr ^ key)
    return np.argsort(h, kind="stable").astype(np.int64)


def _l2n(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.float32)
    n = np.sqrt((x * x).sum(axis=1, dtype=np.float32)).astype(np.float32)
    return (x / n[:, None]).astype(np.float32)


def make_pair(n: int, seed: int, canvas=None, pos_noise=0.5, desc_noise=0.03, outlier_frac=0.1):
    """One synthetic image pair in the reference's GMatcher input layout.

    Returns a dict of NumPy arrays:
      keypoints0/1 (1,N,2) f32, descriptors0/1 (1,256,N) f32, scores0/1 (1,N) f32,
      image0/1 zero uint8 (1,H,W,3) (only .shape is consumed -- gmatcher.py:28,265),
      gt_perm (N,) int64: keypoint i of image0 corresponds to keypoint gt_perm[i]
      of image1, or -1 when its partner was replaced by an outlier.

    ``outlier_frac`` of the image-1 keypoints are replaced by fresh random keypoints with
    unrelated descriptors, so a realistic share of rows/columns must end in the dustbin.
    """
    w, h = canvas if canvas is not None else canvas_for(n)
    xy0 = np.stack([uniform(seed, 1, n) * w, uniform(seed, 2, n) * h], axis=1).astype(np.float32)
    d0 = _l2n(normal(seed, 3, n * 128).reshape(n, 128))
    s0 = uniform(seed, 4, n).astype(np.float32)

    perm = permutation(seed, 5, n)                    # image1[j] = image0[perm[j]]
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n, dtype=np.int64)
    xy1 = xy0[perm].astype(np.float64) + pos_noise * normal(seed, 6, 2 * n).reshape(n, 2)
    xy1[:, 0] = np.clip(xy1[:, 0], 0.0, float(w))
    xy1[:, 1] = np.clip(xy1[:, 1], 0.0, float(h))
    xy1 = xy1.astype(np.float32)
    d1 = d0[perm].astype(np.float64) + desc_noise * normal(seed, 7, n * 128).reshape(n, 128)
    d1 = _l2n(d1)
    s1 = uniform(seed, 8, n).astype(np.float32)
    n_out = int(n * outlier_frac)
    if n_out > 0:
        sel = permutation(seed, 9, n)[:n_out]                     # image-1 slots that become outliers
        xy1[sel] = np.stack([uniform(seed, 10, n_out) * w, uniform(seed, 11, n_out) * h], axis=1).astype(np.float32)
        d1[sel] = _l2n(normal(seed, 12, n_out * 128).reshape(n_out, 128))
        inv[perm[sel]] = -1

    def dup(d):  # (N,128) -> (1,256,N)
        return np.ascontiguousarray(np.concatenate([d, d], axis=1).T[None]).astype(np.float32)

    return {
        "keypoints0": xy0[None].copy(), "keypoints1": xy1[None].copy(),
        "descriptors0": dup(d0), "descriptors1": dup(d1),
        "scores0": s0[None].copy(), "scores1": s1[None].copy(),
        "image0": np.zeros((1, h, w, 3), dtype=np.uint8),
        "image1": np.zeros((1, h, w, 3), dtype=np.uint8),
        "gt_perm": inv,
    }


def make_pair_unbalanced(n0: int, n1: int, n_common: int, seed: int, canvas=None, pos_noise=0.5, desc_noise=0.03):
    """An UNBALANCED pair (n0 != n1) in the same layout as make_pair: image 1 holds the partners of ``n_common`` of image 0's keypoints
    (position and descriptor noise as in make_pair) plus ``n1 - n_common`` fresh outliers, in a shuffled order; the other
    ``n0 - n_common`` keypoints of image 0 have no partner.  The shape of a real pair: the reference's README run has 15 382 / 14 870
    keypoints (README.md:143-163).  gt_perm[i] = index in image 1 of keypoint i's partner, or -1."""
    assert 0 <= n_common <= min(n0, n1)
    w, h = canvas if canvas is not None else canvas_for(max(n0, n1))
    xy0 = np.stack([uniform(seed, 1, n0) * w, uniform(seed, 2, n0) * h], axis=1).astype(np.float32)
    d0 = _l2n(normal(seed, 3, n0 * 128).reshape(n0, 128))
    s0 = uniform(seed, 4, n0).astype(np.float32)
    common = permutation(seed, 5, n0)[:n_common]               # image-0 keypoints that reappear
    slots = permutation(seed, 9, n1)                           # image-1 slots: the first n_common take the partners, the rest outliers
    xy1 = np.empty((n1, 2), dtype=np.float64)
    d1 = np.empty((n1, 128), dtype=np.float64)
    xy1[slots[:n_common]] = xy0[common].astype(np.float64) + pos_noise * normal(seed, 6, 2 * n_common).reshape(n_common, 2)
    d1[slots[:n_common]] = d0[common].astype(np.float64) + desc_noise * normal(seed, 7, n_common * 128).reshape(n_common, 128)
    n_out = n1 - n_common
    xy1[slots[n_common:]] = np.stack([uniform(seed, 10, n_out) * w, uniform(seed, 11, n_out) * h], axis=1)
    d1[slots[n_common:]] = normal(seed, 12, n_out * 128).reshape(n_out, 128)
    xy1[:, 0] = np.clip(xy1[:, 0], 0.0, float(w))
    xy1[:, 1] = np.clip(xy1[:, 1], 0.0, float(h))
    xy1 = xy1.astype(np.float32)
    d1 = _l2n(d1)
    s1 = uniform(seed, 8, n1).astype(np.float32)
    gt = -np.ones(n0, dtype=np.int64)
    gt[common] = slots[:n_common]

    def dup(d):  # (N,128) -> (1,256,N)
        return np.ascontiguousarray(np.concatenate([d, d], axis=1).T[None]).astype(np.float32)

    return {
        "keypoints0": xy0[None].copy(), "keypoints1": xy1[None].copy(),
        "descriptors0": dup(d0), "descriptors1": dup(d1),
        "scores0": s0[None].copy(), "scores1": s1[None].copy(),
        "image0": np.zeros((1, h, w, 3), dtype=np.uint8),
        "image1": np.zeros((1, h, w, 3), dtype=np.uint8),
        "gt_perm": gt,
    }


# ----------------------------------------------------------------------------- weights

GAINS = {"kenc": 0.5, "gnn_encoder": 3.0, "gnn": 0.3, "final_proj": 1.0}


def _gain_for(name: str, gains=None) -> float:
    """Gain of a weight tensor.  ``gains`` may override the module gains (keys of GAINS) and may add substring keys such as
    ``"attn.proj.0"`` (query projection): the longest matching substring key wins over the module gain."""
    g = {**GAINS, **(gains or {})}
    sub = [k for k in g if k not in GAINS and k in name]
    if sub:
        return g[max(sub, key=len)]
    for k in ("gnn_encoder", "kenc", "final_proj", "gnn"):
        if name.startswith(k + "."):
            return g[k]
    return 1.0


def state_dict_spec(descriptor_dim=256, keypoint_encoder=(32, 64, 128, 256), n_layers=18,
                    sage_bias_layout="fc_self", use_layernorm=False):
    """Ordered (name, shape) list mirroring GMatcher.state_dict() (gmatcher.py:177-207).

    Key names follow the reference module tree: ``bin_score``, ``kenc.encoder.*``,
    ``gnn.layers.{i}.attn.{merge,proj.{0,1,2}}``, ``gnn.layers.{i}.mlp.{0,1,3}``,
    ``gnn_encoder.layers.{i}.{fc_neigh,fc_self}``, ``final_proj``.
    """
    D = descriptor_dim
    spec = [("bin_score", ())]
    ch = [2] + list(keypoint_encoder) + [D]
    idx = 0
    for i in range(1, len(ch)):
        spec += [(f"kenc.encoder.{idx}.weight", (ch[i], ch[i - 1], 1)), (f"kenc.encoder.{idx}.bias", (ch[i],))]
        idx += 1
        if i < len(ch) - 1:
            if use_layernorm:                      # gmatcher.py:19-20, 78-79
                spec += [(f"kenc.encoder.{idx}.a_2", (ch[i],)), (f"kenc.encoder.{idx}.b_2", (ch[i],))]
            else:
                for nm in ("weight", "bias", "running_mean", "running_var"):
                    spec.append((f"kenc.encoder.{idx}.{nm}", (ch[i],)))
                spec.append((f"kenc.encoder.{idx}.num_batches_tracked", ()))
            idx += 2  # norm + ReLU
    for l in range(n_layers):
        p = f"gnn.layers.{l}."
        spec += [(p + "attn.merge.weight", (D, D, 1)), (p + "attn.merge.bias", (D,))]
        for j in range(3):
            spec += [(p + f"attn.proj.{j}.weight", (D, D, 1)), (p + f"attn.proj.{j}.bias", (D,))]
        spec += [(p + "mlp.0.weight", (2 * D, 2 * D, 1)), (p + "mlp.0.bias", (2 * D,))]
        if use_layernorm:
            spec += [(p + "mlp.1.a_2", (2 * D,)), (p + "mlp.1.b_2", (2 * D,))]
        else:
            for nm in ("weight", "bias", "running_mean", "running_var"):
                spec.append((p + f"mlp.1.{nm}", (2 * D,)))
            spec.append((p + "mlp.1.num_batches_tracked", ()))
        spec += [(p + "mlp.3.weight", (D, 2 * D, 1)), (p + "mlp.3.bias", (D,))]
    dims = [(D, D // 2), (D // 2, D // 2), (D // 2, D)]
    for i, (ci, co) in enumerate(dims):
        p = f"gnn_encoder.layers.{i}."
        if sage_bias_layout == "fc_self":      # DGL >= 1.0
            spec += [(p + "fc_neigh.weight", (co, ci)), (p + "fc_self.weight", (co, ci)), (p + "fc_self.bias", (co,))]
        else:                                  # older DGL: separate bias parameter
            spec += [(p + "bias", (co,)), (p + "fc_neigh.weight", (co, ci)), (p + "fc_self.weight", (co, ci))]
    spec += [("final_proj.weight", (D, D, 1)), ("final_proj.bias", (D,))]
    return spec


def make_state_dict(seed: int = 123, bias_std: float = 0.02, bn_jitter: float = 0.2, gains=None, head_gains=None, num_heads: int = 4, **kw):
    """Synthetic GMatcher weights as {name: np.ndarray}.

    Conv/linear weights ~ N(0, g^2/fan_in); biases ~ N(0, bias_std^2) (non-zero so the bias
    paths are exercised); BatchNorm gamma/var ~ 1 +- bn_jitter, beta/mean small -- so the
    BN-folding path is exercised as well.  ``bin_score`` = 1 (gmatcher.py:206).  ``gains`` overrides / extends GAINS (see
    ``_gain_for``): e.g. ``{"attn.proj.0": 1.2, "attn.proj.1": 1.2}`` sharpens the attention of every layer.
    ``head_gains``: ``{(layer, head): g}`` multiplies the query and key projection rows of ONE head of one attentional layer (the
    reference interleaves heads: output channel c belongs to head c % num_heads, gmatcher.py:111) by g, bias included -- that
    head's logits grow by g^2 while the other heads of the layer keep theirs.
    """
    out = {}
    for i, (name, shape) in enumerate(state_dict_spec(**kw)):
        n = int(np.prod(shape)) if len(shape) else 1
        stream = 1000 + i
        if name == "bin_score":
            a = np.array(1.0, dtype=np.float32)
        elif name.endswith("num_batches_tracked"):
            a = np.array(0, dtype=np.int64)
        elif name.endswith(".a_2"):
            a = (1.0 + bn_jitter * (2.0 * uniform(seed, stream, n) - 1.0)).astype(np.float32).reshape(shape)
        elif name.endswith(".b_2"):
            a = (bias_std * normal(seed, stream, n)).astype(np.float32).reshape(shape)
        elif name.endswith("running_var"):
            a = (1.0 + bn_jitter * (2.0 * uniform(seed, stream, n) - 1.0)).astype(np.float32).reshape(shape)
        elif name.endswith("running_mean"):
            a = (bias_std * normal(seed, stream, n)).astype(np.float32).reshape(shape)
        elif ".mlp.1." in name or (name.startswith("kenc.encoder.") and len(shape) == 1
                                   and int(name.split(".")[2]) in (1, 4, 7, 10)):
            # BatchNorm affine
            if name.endswith("weight"):
                a = (1.0 + bn_jitter * (2.0 * uniform(seed, stream, n) - 1.0)).astype(np.float32).reshape(shape)
            else:
                a = (bias_std * normal(seed, stream, n)).astype(np.float32).reshape(shape)
        elif name.endswith("bias"):
            a = (bias_std * normal(seed, stream, n)).astype(np.float32).reshape(shape)
        else:
            fan_in = shape[1]
            g = _gain_for(name, gains)
            a = (normal(seed, stream, n) * (g / np.sqrt(float(fan_in)))).astype(np.float32).reshape(shape)
        out[name] = a
    for (layer, head), g in (head_gains or {}).items():
        for j in (0, 1):
            for part in ("weight", "bias"):
                t = out[f"gnn.layers.{layer}.attn.proj.{j}.{part}"]
                t[head::num_heads] = (t[head::num_heads] * np.float32(g)).astype(np.float32)
    return out


# ------------------------------------------------------------------------------------------------ homography pairs
def make_homography(seed: int, canvas, strength: float = 1.0) -> np.ndarray:
    """A mild random homography (float32 3x3, H[2,2] = 1) of a (width, height) canvas around its centre: rotation up to
    +-15 deg, scale 0.9-1.1, shear and perspective terms small enough that every point keeps a positive w.  Portable
    (same hash generator as everything else here), so fixtures only store the seed."""
    w, h = float(canvas[0]), float(canvas[1])
    r = uniform(seed, 77, 8).astype(np.float64)
    ang = np.deg2rad(15.0) * (2 * r[0] - 1) * strength
    sc = 1.0 + 0.1 * (2 * r[1] - 1) * strength
    shx = 0.05 * (2 * r[2] - 1) * strength
    tx, ty = 0.05 * w * (2 * r[3] - 1) * strength, 0.05 * h * (2 * r[4] - 1) * strength
    px, py = 1e-4 * (2 * r[5] - 1) * strength, 1e-4 * (2 * r[6] - 1) * strength
    c, s = np.cos(ang), np.sin(ang)
    A = np.array([[sc * c, -sc * s + shx, 0.0], [sc * s, sc * c, 0.0], [0.0, 0.0, 1.0]])
    T0 = np.array([[1, 0, -w / 2], [0, 1, -h / 2], [0, 0, 1.0]])
    T1 = np.array([[1, 0, w / 2 + tx], [0, 1, h / 2 + ty], [0, 0, 1.0]])
    Pm = np.array([[1, 0, 0], [0, 1, 0], [px, py, 1.0]])
    H = T1 @ Pm @ A @ T0
    return (H / H[2, 2]).astype(np.float32)


def make_homography_pair(n: int, seed: int, canvas=None, pos_noise: float = 0.5, desc_noise: float = 0.03, outlier_frac: float = 0.1):
    """Like make_pair, but image 1's keypoints are image 0's warped by a random homography (then permuted, jittered by
    `pos_noise` px, with `outlier_frac` of them replaced): the ground truth the reference's eval loop is given as
    `homo_matrix` (eval_homography.py:164-165).  Returns (pair dict, H float32 3x3)."""
    pair = make_pair(n, seed, canvas=canvas, pos_noise=0.0, desc_noise=desc_noise, outlier_frac=outlier_frac)
    cv = canvas_for(n) if canvas is None else canvas
    H = make_homography(seed, cv)
    gt = pair["gt_perm"]
    k0 = pair["keypoints0"][0].astype(np.float64)
    src = np.concatenate([k0, np.ones((len(k0), 1))], axis=1)
    dst = (H.astype(np.float64) @ src.T).T
    dst = dst[:, :2] / dst[:, 2:3]
    k1 = pair["keypoints1"][0].copy()
    jitter = normal(seed, 78, 2 * len(k0)).reshape(len(k0), 2) * pos_noise
    ok = gt >= 0
    k1[gt[ok]] = (dst[ok] + jitter[ok]).astype(np.float32)
    pair["keypoints1"] = k1[None]
    return pair, H


# ------------------------------------------------------------------------------------------------ CAR-HyNet
def carhynet_state_dict_spec():
    """(name, shape) of every tensor of the reference's CAR_HyNet().state_dict() (carhynet/models.py:311-362), in its order."""
    spec = []

    def frn(p, c):
        spec.extend([(p + "weight", (1, c, 1, 1)), (p + "bias", (1, c, 1, 1)), (p + "eps", (1,))])

    def bn(p, c, affine=True):
        if affine:
            spec.extend([(p + "weight", (c,)), (p + "bias", (c,))])
        spec.extend([(p + "running_mean", (c,)), (p + "running_var", (c,)), (p + "num_batches_tracked", ())])

    def coordatt(p, c, mip=8):
        spec.extend([(p + "conv1.weight", (mip, c, 1, 1)), (p + "conv1.bias", (mip,))])
        bn(p + "bn1.", mip)
        spec.extend([(p + "conv_h.weight", (c, mip, 1, 1)), (p + "conv_h.bias", (c,)), (p + "conv_w.weight", (c, mip, 1, 1)), (p + "conv_w.bias", (c,))])

    def sandglass(p, c, hidden=16):
        spec.append((p + "conv.0.0.weight", (c, 1, 3, 3))); bn(p + "conv.0.1.", c)
        coordatt(p + "conv.1.", c)
        spec.append((p + "conv.2.weight", (hidden, c, 1, 1))); bn(p + "conv.3.", hidden)
        spec.append((p + "conv.4.0.weight", (c, hidden, 1, 1))); bn(p + "conv.4.1.", c)
        spec.append((p + "conv.5.weight", (c, 1, 3, 3))); bn(p + "conv.6.", c)

    frn("layer1.0.", 3); spec.append(("layer1.1.tau", (1, 3, 1, 1)))
    spec.extend([("layer1.2.weight", (32, 3, 3, 3)), ("layer1.2.bias", (32,))]); frn("layer1.3.", 32); coordatt("layer1.4.", 32)
    spec.append(("layer1.5.tau", (1, 32, 1, 1)))
    spec.extend([("layer2.0.weight", (32, 32, 3, 3)), ("layer2.0.bias", (32,))]); frn("layer2.1.", 32); coordatt("layer2.2.", 32)
    spec.append(("layer2.3.tau", (1, 32, 1, 1)))
    sandglass("layer2_5.", 32)
    for name, cin, cout in (("layer3", 32, 64), ("layer4", 64, 64)):
        spec.extend([(name + ".0.weight", (cout, cin, 3, 3)), (name + ".0.bias", (cout,))]); frn(name + ".1.", cout)
        spec.append((name + ".2.tau", (1, cout, 1, 1)))
        if name == "layer4":
            sandglass("layer4_5.", 64)
    for name, cin, cout in (("layer5", 64, 128), ("layer6", 128, 128)):
        spec.extend([(name + ".0.weight", (cout, cin, 3, 3)), (name + ".0.bias", (cout,))]); frn(name + ".1.", cout)
        spec.append((name + ".2.tau", (1, cout, 1, 1)))
    spec.append(("layer7.1.weight", (128, 128, 8, 8))); bn("layer7.2.", 128, affine=False)
    return spec


def make_carhynet_state_dict(seed: int = 321, jitter: float = 0.2, bias_std: float = 0.05):
    """Synthetic CAR-HyNet weights as {name: np.ndarray} (the reference's ./weights/car_hynet.pth is not in the repository).
    Conv weights ~ N(0, 2 / fan_in) (activations keep their scale through the FRN / TLU stack), biases and BatchNorm / FRN
    shifts ~ N(0, bias_std^2), scales and variances 1 +- jitter, TLU thresholds around the reference's init -1 (models.py:101),
    FRN eps = 1e-6 (models.py:24)."""
    out = {}
    for i, (name, shape) in enumerate(carhynet_state_dict_spec()):
        n = int(np.prod(shape)) if len(shape) else 1
        stream = 5000 + i
        if name.endswith("num_batches_tracked"):
            a = np.array(0, dtype=np.int64)
        elif name.endswith(".eps"):
            a = np.array([1e-6], dtype=np.float32)
        elif name.endswith(".tau"):
            a = (-1.0 + 0.5 * (2.0 * uniform(seed, stream, n) - 1.0)).astype(np.float32).reshape(shape)
        elif name.endswith("running_var") or (name.endswith("weight") and (len(shape) == 1 or shape[0] == 1)):
            a = (1.0 + jitter * (2.0 * uniform(seed, stream, n) - 1.0)).astype(np.float32).reshape(shape)     # BN gamma / var, FRN scale
        elif name.endswith("running_mean") or name.endswith("bias"):
            a = (bias_std * normal(seed, stream, n)).astype(np.float32).reshape(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = (normal(seed, stream, n) * np.sqrt(2.0 / fan_in)).astype(np.float32).reshape(shape)
        out[name] = a
    return out


def make_patches(n: int, seed: int):
    """n synthetic 32x32x3 patches in [0, 1] (NHWC float32, what HyNetnetFeature2D.compute_des_batches takes, models.py:655-666):
    a smooth random field per patch (low-frequency sinusoids) plus noise, so neighbouring pixels correlate like image patches."""
    yy, xx = np.meshgrid(np.arange(32, dtype=np.float32), np.arange(32, dtype=np.float32), indexing="ij")
    u = uniform(seed, 77, n * 3 * 6).reshape(n, 3, 6).astype(np.float32)
    noise = normal(seed, 78, n * 32 * 32 * 3).reshape(n, 32, 32, 3).astype(np.float32)
    out = np.empty((n, 32, 32, 3), dtype=np.float32)
    for c in range(3):
        fx, fy, ph = 0.05 + 0.3 * u[:, c, 0], 0.05 + 0.3 * u[:, c, 1], 6.2831853 * u[:, c, 2]
        amp, off = 0.2 + 0.25 * u[:, c, 3], 0.3 + 0.4 * u[:, c, 4]
        field = off[:, None, None] + amp[:, None, None] * np.sin(fx[:, None, None] * xx[None] + fy[:, None, None] * yy[None] + ph[:, None, None])
        out[..., c] = field
    return np.clip(out + 0.03 * noise, 0.0, 1.0).astype(np.float32)

"""Device-side front end: patch extraction + CAR-HyNet descriptors on the GPU (SURVEY 8f, rows f4 + f1).

The reference's ``utils.common.sift_forward`` (utils/common.py:837-893) does, per image: OpenCV SIFT detection -> Gaussian
pyramid + one affine-warped 64x64 patch per keypoint (utils/library.py:84-110, 234-271, one CPU core: 3.2-3.9 s per image
at 15 k keypoints, README.md:151,155) -> CAR-HyNet on 32x32 patches -> 128-d descriptors duplicated to 256-d.  Here
everything after the detection runs on the MI355X: ``extract_patches`` (csrc/patches.hip) writes the patches straight in
the NHWC float layout the CAR-HyNet kernels read, so they never exist on the host.

``sift_forward_device`` has ``sift_forward``'s signature and can be handed to ``Matching`` as ``config['front_end']``.
Keypoint DETECTION stays what the caller provides: ``detector(img) -> keypoints`` (objects with cv2.KeyPoint's attributes
``pt, size, angle, response, octave``, or a structured array with those fields); by default OpenCV's SIFT with the
reference's parameters when ``cv2`` is importable.  The patch arithmetic restates OpenCV's uint8 fixed-point resampling;
it is PARITY UNPINNED against OpenCV itself (see csrc/patches.hip and DESIGN.md).
"""
from __future__ import annotations

import numpy as np
import torch

from . import hip


def keypoint_arrays(kps):
    """cv2.KeyPoint-like objects (or a structured array / dict of arrays) -> (kp4 float32 [n, 4] = x, y, size, angle;
    octave int32 [n]; response float32 [n])."""
    if isinstance(kps, dict):
        pt = np.asarray(kps["pt"], dtype=np.float32).reshape(-1, 2)
        kp4 = np.concatenate([pt, np.asarray(kps["size"], np.float32)[:, None], np.asarray(kps["angle"], np.float32)[:, None]], 1)
        return np.ascontiguousarray(kp4), np.asarray(kps["octave"], np.int32), np.asarray(kps.get("response", np.zeros(len(pt))), np.float32)
    n = len(kps)
    kp4 = np.empty((n, 4), dtype=np.float32)
    octv = np.empty(n, dtype=np.int32)
    resp = np.empty(n, dtype=np.float32)
    for i, k in enumerate(kps):
        kp4[i] = (k.pt[0], k.pt[1], k.size, k.angle)
        octv[i] = k.octave
        resp[i] = k.response
    return kp4, octv, resp


def extract_patches(img, kps, device=None):
    """buildGaussianPyramid + ComputePatches(radius_size=64) + INTER_AREA to 32x32 + / 255 (utils/common.py:882-884) on the
    device.  img: uint8 [H, W, 3] (NumPy or tensor); kps: see keypoint_arrays.  Returns float32 [n, 32, 32, 3] on the device.
    Keypoints whose octave / layer fall outside the pyramid raise IndexError, like the reference's list indexing would."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    t = torch.as_tensor(np.ascontiguousarray(img) if isinstance(img, np.ndarray) else img)
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
        raise ValueError("extract_patches takes a uint8 [H, W, 3] image (what sift_forward passes with graydesc=False)")
    t = t.to(dev).contiguous()
    kp4, octv, _ = keypoint_arrays(kps)
    pyr, levels, dev_levels = hip.pyramid_build(t)
    out, bad = hip.patch_extract(pyr, dev_levels, len(levels), torch.from_numpy(kp4).to(dev), torch.from_numpy(octv).to(dev))
    if len(kp4) and int(bad.item()):
        raise IndexError(f"{int(bad.item())} keypoint(s) reference a pyramid level that does not exist (octave / layer out of range)")
    return out


def filter_max_num(kps, max_num):
    """filterMaxNumDesc (utils/common.py:710-718): the max_num keypoints with the largest response, in descending order."""
    if 0 < max_num < len(kps):
        responses = [k.response for k in kps]
        idxs = np.fliplr(np.reshape(np.argsort(responses), (1, -1))).reshape(-1)
        return [kps[idxs[n]] for n in range(max_num)]
    return kps


class PaddedKeyPoint:
    """What ``cv2.KeyPoint(x, y, 1)`` carries (utils/common.py:683): size 1, angle -1, response 0, octave 0, class_id -1."""
    __slots__ = ("pt", "size", "angle", "response", "octave", "class_id")

    def __init__(self, x, y):
        self.pt, self.size, self.angle, self.response, self.octave, self.class_id = (float(x), float(y)), 1.0, -1.0, 0.0, 0, -1


def pad_training_keypoints(kps, max_keypoints, img_shape):
    """The ``is_train`` branch of sift_forward (utils/common.py:866-880, ``copy=False``): an image with fewer than max_keypoints
    detections gets random extra locations -- x and y drawn with the reference's own sequence of ``np.random`` calls (a
    (n, 2) block scaled by the width, then the y column redrawn and scaled by the height), so a seeded run places them where
    the reference does -- as size-1 keypoints, so that every image of a training batch carries exactly max_keypoints
    (train.py:107-110 stacks them).  The reference passes the locations through ``cv2.SIFT.compute`` (682-685), which may
    drop points OpenCV considers too close to the border: NOT restated (OpenCV is not available here; parity unpinned)."""
    kps = list(kps)
    if max_keypoints <= 0 or len(kps) >= max_keypoints:
        return kps
    to_add = max_keypoints - len(kps)
    coordinates = np.random.random((to_add, 2)) * img_shape[1]
    coordinates[:, 1] = np.random.random(to_add) * img_shape[0]
    return kps + [PaddedKeyPoint(x, y) for x, y in coordinates]


def _default_detector():
    try:
        import cv2
    except Exception as e:             # no OpenCV in this environment
        raise NotImplementedError("keypoint DETECTION needs OpenCV SIFT (cv2 is not importable here): pass detector=callable(img) -> "
                                  "keypoints (cv2.KeyPoint-like objects)") from e
    sift = cv2.SIFT_create(nfeatures=None, nOctaveLayers=3, contrastThreshold=0.001, edgeThreshold=80, sigma=1.6)   # common.py:838-857
    return lambda img: sift.detect(img, None)


def sift_forward_device(data, device, detector=None):
    """``utils.common.sift_forward`` (common.py:837-893; ``data['is_train']``: pad_training_keypoints) with patches and descriptors on the device.
    data: {'image': uint8 [B, H, W, 3], 'max_keypoints': int, 'carhynet': gims_amd.carhynet.CARHyNet (or any object with
    ``_forward_nhwc`` / ``compute_des_batches``)}.  Returns {'keypoints', 'scores', 'descriptors'}: lists with one tensor per
    image -- (n, 2), (n,), (256, n) with the 128-d descriptor duplicated (common.py:890-892)."""
    det = detector or data.get("detector") or _default_detector()
    net = data["carhynet"]
    kpts, descs, scores = [], [], []
    for img in data["image"]:
        img = np.asarray(img.cpu() if torch.is_tensor(img) else img)
        k = filter_max_num(list(det(img)), data.get("max_keypoints", -1))
        if data.get("is_train", False):
            k = pad_training_keypoints(k, data.get("max_keypoints", -1), img.shape)
        kp4, _, resp = keypoint_arrays(k)
        patches = extract_patches(img, k, device)
        if hasattr(net, "_forward_nhwc"):                     # gims_amd.carhynet.CARHyNet: stays on the device
            with torch.no_grad():
                d = net._forward_nhwc(patches)[0]
        else:                                                 # the reference's HyNetnetFeature2D: NumPy in, NumPy out
            d = torch.from_numpy(np.asarray(net.compute_des_batches(patches.cpu().numpy(), True), dtype=np.float32)).to(device)
        kpts.append(torch.from_numpy(np.ascontiguousarray(kp4[:, :2])).to(device))
        descs.append(torch.cat([d, d], dim=1).permute(1, 0).to(device))
        scores.append(torch.from_numpy(resp).to(device))
    return {"keypoints": kpts, "scores": scores, "descriptors": descs}

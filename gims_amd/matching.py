"""``Matching`` front-end shell with the reference's signature and behaviour (models/matching.py:8-30).

When the caller does not supply keypoints (the way ``eval_homography.py:177`` and ``eval_matches.py:152`` call it:
``image0/image1/carhynet/device`` only) the reference runs ``utils.common.sift_forward`` per image (OpenCV SIFT
detection + patch extraction + the CAR-HyNet CNN, utils/common.py:837-893) and merges its outputs into the returned
dict.  This shell does the same through a FRONT END, looked up in this order:

  1. ``config['front_end']`` -- a callable with ``sift_forward``'s signature
     ``fe({'image': ..., 'max_keypoints': ..., 'carhynet': ...}, device=...) -> {'keypoints': [..], 'scores': [..],
     'descriptors': [..]}`` (lists with one tensor per image of the batch);
  2. the caller-side ``utils.common.sift_forward`` when it is importable (running inside the reference's tree: the
     swap in eval_homography.py is then the import line only);
  3. otherwise ``NotImplementedError`` naming what is missing (OpenCV SIFT is a CPU stage outside this hot path).

``data['carhynet']`` may be the reference's ``HyNetnetFeature2D`` or ``gims_amd.carhynet.CARHyNet`` (same
``compute_sift`` / ``compute_des_batches`` methods, descriptors computed by the HIP kernels).
"""
import torch

from .gmatcher import GMatcher


def _find_front_end(config):
    fe = config.get('front_end')
    if fe is not None:
        if not callable(fe):
            raise TypeError("config['front_end'] must be callable like utils.common.sift_forward(data, device)")
        return fe
    try:
        from utils.common import sift_forward          # the reference's own front end, when running inside its tree
        return sift_forward
    except Exception:                                   # not importable here (no OpenCV, or not in the reference's tree)
        return None


class Matching(torch.nn.Module):
    """ Image Matching Frontend """

    def __init__(self, config={}):
        super().__init__()
        self.gmodel = GMatcher({k: v for k, v in config.items() if k != 'front_end'})
        self.max_keypoints = config.get('max_keypoints', -1)
        self._front_end = _find_front_end(config)

    def forward(self, data):
        pred = {}
        for side in ('0', '1'):                         # models/matching.py:17-24
            if 'keypoints' + side in data:
                continue
            if self._front_end is None:
                raise NotImplementedError(
                    f"no 'keypoints{side}' in data and no front end: pass config['front_end'] (a callable like "
                    "utils.common.sift_forward) or run where `utils.common.sift_forward` is importable; OpenCV SIFT "
                    "detection is a CPU stage outside the MI355X hot path")
            p = self._front_end({'image': data['image' + side], 'max_keypoints': self.max_keypoints,
                                 'carhynet': data.get('carhynet')}, device=data['device'])
            pred = {**pred, **{k + side: v for k, v in p.items()}}
        data = {**data, **pred}
        for k in data:
            if isinstance(data[k], (list, tuple)):
                data[k] = torch.stack(data[k])
        pred = {**pred, **self.gmodel(data)}
        return pred

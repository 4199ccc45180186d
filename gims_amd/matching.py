"""``Matching`` front-end shell with the reference's signature (models/matching.py:8-30).

The reference runs OpenCV SIFT + patch extraction + the CAR-HyNet CNN here when the caller does not
supply keypoints (utils/common.py:837-893).  That CPU/OpenCV front end is outside this hot path
(SURVEY.md section 8, rows f1/f4): callers hand in ``keypoints0/1``, ``descriptors0/1``, ``scores0/1``
-- exactly the tensors ``sift_forward`` would have produced -- and this shell does the list->tensor
stacking (matching.py:26-28) and calls ``GMatcher``.
"""
import torch

from .gmatcher import GMatcher


class Matching(torch.nn.Module):
    """ Image Matching Frontend """

    def __init__(self, config={}):
        super().__init__()
        self.gmodel = GMatcher(config)
        self.max_keypoints = config.get('max_keypoints', -1)

    def forward(self, data):
        missing = [k for k in ('keypoints0', 'keypoints1') if k not in data]
        if missing:
            raise NotImplementedError(
                "SIFT + CAR-HyNet keypoint extraction (utils/common.py:837-893) is outside the MI355X hot path; "
                f"pass {missing} / descriptors / scores in `data`")
        data = {**data}
        for k in data:
            if isinstance(data[k], (list, tuple)):
                data[k] = torch.stack(data[k])
        return {**self.gmodel(data)}

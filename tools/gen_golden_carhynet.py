#!/usr/bin/env python3
"""Golden vectors for the CAR-HyNet descriptor, produced by the REFERENCE itself in the build container.

Imports /root/reference/carhynet/models.py (cv2, which carhynet/util.py imports only for reading image files, is stubbed),
loads the portable synthetic weights of gims_amd.synth.make_carhynet_state_dict into the reference's CAR_HyNet, runs
seeded synthetic patches through it in eval mode and stores seeds + outputs in tests/golden/carhynet_*.npz.
Run here only (the reference does not travel to the GPU box):  python tools/gen_golden_carhynet.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, "/root/reference")
from carhynet.models import CAR_HyNet  # noqa: E402
from gims_amd import synth  # noqa: E402


def main():
    torch.manual_seed(0)
    model = CAR_HyNet().eval()
    spec = synth.carhynet_state_dict_spec()
    ref_keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    assert ref_keys == [(k, tuple(s)) for k, s in spec], "gims_amd.synth.carhynet_state_dict_spec no longer mirrors the reference"
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, seed_w, seed_p, n in (("carhynet_n24_w321_p5", 321, 5, 24), ("carhynet_n3_w322_p9", 322, 9, 3)):
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_carhynet_state_dict(seed_w).items()}
        model.load_state_dict(sd)
        patches = torch.from_numpy(synth.make_patches(n, seed_p))
        with torch.no_grad():
            desc, raw = model(patches.permute(0, 3, 1, 2), mode="train")      # (L2-normalised, raw) -- eval-mode modules, models.py:396-397
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), seed_w=seed_w, seed_p=seed_p, n=n,
                            desc=desc.numpy().astype(np.float32), raw=raw.numpy().astype(np.float32),
                            n_params=sum(int(np.prod(s)) if len(s) else 1 for _, s in spec))
        print(name, "desc", tuple(desc.shape), "max|raw|", float(raw.abs().max()))


if __name__ == "__main__":
    main()

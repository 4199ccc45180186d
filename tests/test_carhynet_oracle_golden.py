"""CPU: the CAR-HyNet oracle against golden vectors produced by the reference itself (tools/gen_golden_carhynet.py)."""
import glob
import os

import numpy as np
import pytest
import torch

from gims_amd import synth
from oracle import carhynet_oracle as CO

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "carhynet_*.npz")))


def test_goldens_present():
    assert len(GOLD) >= 2


def test_state_dict_spec_matches_reference_inventory():
    spec = synth.carhynet_state_dict_spec()
    assert len(spec) == 136                                           # tensors of CAR_HyNet().state_dict()
    n_params = sum(int(np.prod(s)) if len(s) else 1 for _, s in spec)
    assert n_params == 1347613
    for g in GOLD:
        assert int(np.load(g)["n_params"]) == n_params


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(g)[:-4] for g in GOLD])
def test_oracle_matches_reference(path):
    g = np.load(path)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_carhynet_state_dict(int(g["seed_w"])).items()}
    patches = torch.from_numpy(synth.make_patches(int(g["n"]), int(g["seed_p"])))
    with torch.no_grad():
        desc, raw = CO.car_hynet_forward(sd, patches)
    # same f32 arithmetic, BatchNorm written out instead of F.batch_norm: reorder noise only
    np.testing.assert_allclose(raw.numpy(), g["raw"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(desc.numpy(), g["desc"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(np.linalg.norm(desc.numpy(), axis=1), 1.0, atol=1e-5)


def test_patch_generator_is_portable():
    p = synth.make_patches(5, 3)
    assert p.shape == (5, 32, 32, 3) and p.dtype == np.float32 and p.min() >= 0.0 and p.max() <= 1.0
    assert abs(float(p.astype(np.float64).sum()) - float(synth.make_patches(5, 3).astype(np.float64).sum())) == 0.0

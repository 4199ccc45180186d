"""The oracle's training step (oracle/gims_oracle.py: train_step -- train-mode BatchNorm, torch autograd through the
restatement) against the reference's own training step (tools/gen_golden_grads.py: gmodel.train(), forward(mode='train'),
loss.backward(); train.py:100, 136-137).  CPU only."""
import numpy as np
import pytest

from gims_amd import synth
from tests.helpers import check_step_gradients, golden_names, load_golden, train_data, train_pairs
from oracle import gims_oracle as O


@pytest.mark.parametrize("name", [n for n in golden_names("trainstep_") if "n2048" not in n and "n4096" not in n])
def test_oracle_train_step_vs_reference(name):
    g = load_golden(name)
    pairs = train_pairs(name, g)
    data = train_data(pairs, g)
    cfg = dict(sinkhorn_iterations=int(g["meta"][4]), pos_loss_weight=float(g["pos_loss_weight"]), neg_loss_weight=float(g["neg_loss_weight"]))
    (loss, pos, neg), grads, bufs = O.train_step(synth.make_state_dict(123, use_layernorm=name.startswith("trainstep_ln_")), data, cfg)
    assert abs(loss - float(g["loss"])) <= 2e-6 * max(1.0, abs(float(g["loss"])))
    assert abs(pos - float(g["pos"])) <= 2e-6 and abs(neg - float(g["neg"])) <= 2e-6
    worst = check_step_gradients(g, grads, rtol=1e-2, rtol_p95=1e-3)
    print(name, "worst gradient error (relative to the tensor's largest entry)", worst)
    for k in g:
        if k.startswith("b:"):
            ref, mine = g[k], bufs[k[2:]]
            if k.endswith("num_batches_tracked"):
                assert int(mine) == int(ref), k
            else:
                np.testing.assert_allclose(mine, ref, rtol=2e-5, atol=2e-6, err_msg=k)

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


def test_relu_flips_are_what_bounds_the_gradient_bar():
    """the worst-tensor bar of the training-step tests (1e-2 here, 2e-2 for the HIP step) is attributed to ReLU decisions that
    fall the other way between two evaluations of the same step.  Isolated here on the CPU: the oracle's step in float32 against the SAME step in
    float64 (a) with every run taking its own ReLU decisions and (b) with the float64 run's decisions imposed on the float32 run.  A handful of the
    five million activations flip, the worst tensor is off by ~2e-3 in (a) -- and by ~2.5e-4 in (b), where the 95th percentile does not move: what
    is left without the flips is ordinary f32 rounding, an order of magnitude under the bar."""
    import torch
    import torch.nn.functional as F
    name = "trainstep_n256_s1002_i100"
    g = load_golden(name)
    pairs = train_pairs(name, g)
    cfg = dict(sinkhorn_iterations=int(g["meta"][4]), pos_loss_weight=float(g["pos_loss_weight"]), neg_loss_weight=float(g["neg_loss_weight"]))
    sd32 = synth.make_state_dict(123)
    sd64 = {k: (np.asarray(v).astype(np.float64) if np.asarray(v).dtype == np.float32 else v) for k, v in sd32.items()}

    def data(dt):
        x = train_data(pairs, g)
        return {k: (v.to(dt) if torch.is_tensor(v) and v.dtype == torch.float32 else v) for k, v in x.items()}

    masks, counts = [], []

    def run(sd, dt, mode):
        idx = [0]

        def relu(x):
            if mode == "record":
                masks.append(x > 0)
                return F.relu(x)
            m = masks[idx[0]]
            idx[0] += 1
            counts.append((int(((x > 0) != m).sum()), x.numel()))
            return x * m.to(x.dtype) if mode == "impose" else F.relu(x)
        shim = type("FShim", (), {"__getattr__": lambda self, n: relu if n == "relu" else getattr(F, n)})()
        old, O.F = O.F, shim
        try:
            return O.train_step(sd, data(dt), cfg)[1]
        finally:
            O.F = old

    def profile(a, b):
        big = max(np.abs(v).max() for v in b.values())
        errs = np.sort([np.abs(a[k] - b[k]).max() / max(np.abs(b[k]).max(), 1e-3 * big) for k in b])
        return float(errs[-1]), float(errs[int(0.95 * (len(errs) - 1))])

    g64 = run(sd64, torch.float64, "record")
    g32 = run(sd32, torch.float32, "own")
    flips, total = sum(f for f, _ in counts), sum(n for _, n in counts)
    g32i = run(sd32, torch.float32, "impose")
    own, imposed = profile(g32, g64), profile(g32i, g64)
    print(f"{len(masks)} ReLU calls, {flips} of {total} activations flip between f32 and f64; worst / p95 gradient difference: own decisions "
          f"{own[0]:.2e} / {own[1]:.2e}, f64's decisions imposed {imposed[0]:.2e} / {imposed[1]:.2e}")
    assert 0 < flips <= 50 and total > 1_000_000
    assert imposed[0] < 6e-4 and own[0] > 3.0 * imposed[0]          # the flips are the worst tensor's error ...
    assert abs(own[1] - imposed[1]) < 0.5 * own[1]                   # ... and nothing else: the bulk of the tensors does not notice them

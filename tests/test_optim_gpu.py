"""gims_amd.optim.Adam (csrc/optim.hip, gims_adam_step) against torch.optim.Adam -- the optimizer train.py:52-57 builds -- on the same
parameters, gradients and parameter groups."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _autograd_on():
    """Other test modules switch autograd off process-wide at import; a training step needs it."""
    with torch.enable_grad():
        yield


def _params(seed, dev, shapes):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.randn(*s, generator=g) * 0.3).to(dev)) for s in shapes]


SHAPES = [(256, 256, 1), (256,), (1,), (3, 5, 7), (4097,), (512, 512, 1), (0,), (33,), (768, 256, 1), (2, 2)]


def _both(dev, wd=1e-4, lr=1e-3):
    from gims_amd.optim import Adam
    pa, pb = _params(3, dev, SHAPES), _params(3, dev, SHAPES)
    oa = torch.optim.Adam(pa[:3], lr=lr, betas=(0.9, 0.999), foreach=False)
    ob = Adam(pb[:3], lr=lr, betas=(0.9, 0.999))
    for o, p in ((oa, pa), (ob, pb)):                 # train.py:56-57
        o.add_param_group({'params': p[3:7], 'weight_decay': wd})
        o.add_param_group({'params': p[7:]})
    return pa, pb, oa, ob


def _set_grads(ps, seed, skip=()):
    g = torch.Generator().manual_seed(seed)
    for i, p in enumerate(ps):
        gr = torch.randn(*p.shape, generator=g) * (10.0 ** float(torch.randint(-4, 2, (1,), generator=g)))
        p.grad = None if i in skip else gr.to(p.device)


def test_adam_matches_torch_over_steps():
    dev = torch.device("cuda:0")
    pa, pb, oa, ob = _both(dev)
    for step in range(1, 8):
        skip = (4,) if step in (3, 4) else ()             # a parameter without a gradient keeps its own step count
        _set_grads(pa, 100 + step, skip)
        _set_grads(pb, 100 + step, skip)
        if step == 5:                                     # train.py:21-26 change_lr
            for o in (oa, ob):
                for g in o.param_groups:
                    g['lr'] = 3e-4
        oa.step()
        ob.step()
        for i, (a, b) in enumerate(zip(pa, pb)):
            # the same float32 operations in the same order; division and square root are correctly rounded on both sides
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-9), (step, i, float((a - b).abs().max()))
            sa, sb = oa.state.get(a, {}), ob.state.get(b, {})
            if sa:
                assert float(sa['step']) == float(sb['step'])
                assert torch.allclose(sa['exp_avg'], sb['exp_avg'], rtol=1e-6, atol=1e-12)
                assert torch.allclose(sa['exp_avg_sq'], sb['exp_avg_sq'], rtol=1e-6, atol=1e-20)
    # bitwise on the typical path (every tensor, first step from zero moments) -- a regression guard, measured equal
    pa2, pb2, oa2, ob2 = _both(dev)
    _set_grads(pa2, 7)
    _set_grads(pb2, 7)
    oa2.step(); ob2.step()
    worst = max(float((a - b).abs().max()) for a, b in zip(pa2, pb2) if a.numel())
    assert worst <= 1e-7, worst


def test_adam_state_dict_round_trips_with_torch():
    dev = torch.device("cuda:0")
    pa, pb, oa, ob = _both(dev)
    for step in range(3):
        _set_grads(pa, 50 + step)
        _set_grads(pb, 50 + step)
        oa.step(); ob.step()
    sd_a, sd_b = oa.state_dict(), ob.state_dict()
    assert set(sd_a['state'].keys()) == set(sd_b['state'].keys())
    for k in sd_a['state']:
        assert set(sd_a['state'][k].keys()) == set(sd_b['state'][k].keys()) == {'step', 'exp_avg', 'exp_avg_sq'}
    # cross-load: ours <- torch's checkpoint and torch's <- ours, then one more step on each side
    ob.load_state_dict(sd_a)
    oa.load_state_dict(sd_b)
    _set_grads(pa, 99)
    _set_grads(pb, 99)
    oa.step(); ob.step()
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=4e-6, atol=1e-9)
    assert float(ob.state[pb[0]]['step']) == 4.0


def test_adam_refuses_what_is_not_built():
    from gims_amd.optim import Adam
    p = [torch.nn.Parameter(torch.zeros(4))]
    with pytest.raises(NotImplementedError):
        Adam(p, amsgrad=True)
    o = Adam(p)
    p[0].grad = torch.ones(4)
    with pytest.raises(RuntimeError):                     # CPU parameter: no CPU path
        o.step()


def test_training_steps_with_the_fused_optimizer_track_torch_adam():
    """The reference's loop (train.py:136-139: forward(mode='train'), backward, optimizer.step(), zero_grad) with gims_amd.optim.Adam
    and with torch.optim.Adam on two copies of one model: same losses step by step."""
    from gims_amd.optim import Adam
    from gims_amd import synth
    from tests.helpers import load_golden, train_data, train_pairs
    from tests.test_trainstep_gpu import _model
    name = "trainstep_n256_s1002_i100"
    g = load_golden(name)
    pairs = train_pairs(name, g)
    losses = []
    for make in (lambda ps: torch.optim.Adam(ps, lr=2e-4, foreach=False), lambda ps: Adam(ps, lr=2e-4)):
        m = _model(synth.make_state_dict(123), g)
        opt = make(m.parameters())
        ls = []
        for _ in range(4):
            loss, _, _ = m(train_data(pairs, g, device="cuda"), mode="train")
            loss.backward()
            opt.step()
            opt.zero_grad()
            ls.append(float(loss))
        losses.append(ls)
    assert losses[0][-1] < losses[0][0]
    assert np.allclose(losses[0], losses[1], rtol=1e-4, atol=1e-6), losses

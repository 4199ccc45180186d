"""Kernels of the training step (gims_amd/csrc/train.hip) against float64 / torch references of the same operations."""
import numpy as np
import pytest
import torch

from gims_amd import hip

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


@pytest.mark.parametrize("m,n,k", [(128, 128, 32), (300, 200, 70), (2031, 64, 2031), (65, 257, 2), (1, 512, 513), (257, 1, 33), (512, 512, 512)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("prec,tol", [(hip.PREC_BF16X3, 1.5e-5), (hip.PREC_BF16X6, 4e-7)])
def test_gemm_f32_forms(m, n, k, ta, tb, prec, tol):
    """C = alpha A B^T + beta C + bias + residual for both storage orders of both operands, ragged edges, K not a multiple of 4."""
    a = _rand(k, m, seed=1).t() if ta else _rand(m, k, seed=1)
    b = _rand(k, n, seed=2).t() if tb else _rand(n, k, seed=2)
    c0 = _rand(m, n, seed=3)
    bias, res = _rand(n, seed=4), _rand(m, n, seed=5)
    out = c0.clone()
    hip.gemm(a, b, out, alpha=0.5, beta=2.0, bias=bias, residual=res, precision=prec)
    ref = 0.5 * (a.double() @ b.double().t()) + 2.0 * c0.double() + bias.double() + res.double()
    err = float((out.double() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    assert err < tol, err            # x3: operands carry 16 mantissa bits; x6: 24 (f32 accumulation over k either way)
    out2 = hip.gemm(a, b, act=hip.ACT_RELU, precision=prec)
    ref2 = (a.double() @ b.double().t()).clamp(min=0)
    assert float((out2.double() - ref2).abs().max()) / max(1.0, float(ref2.abs().max())) < tol


def test_gemm_batched_strided_heads():
    """Batched over heads with column-slice operands: S_h = Q_h K_h^T / 8 and O_h = P_h V_h written into a [rows, 256] buffer."""
    n, mk = 333, 271
    q, kv = _rand(n, 256, seed=1), _rand(mk, 512, seed=2)
    qh = q.view(n, 4, 64).permute(1, 0, 2)                       # [4, n, 64] view, batch stride 64
    kh = kv[:, :256].view(mk, 4, 64).permute(1, 0, 2)
    vh = kv[:, 256:].view(mk, 4, 64).permute(1, 0, 2)
    ld = (mk + 3) // 4 * 4
    s = torch.zeros(4, n, ld, device=DEV)
    hip.gemm(qh, kh, s[:, :, :mk], alpha=0.125)
    ref = torch.einsum("hnd,hmd->hnm", qh.double(), kh.double()) / 8
    assert float((s[:, :, :mk].double() - ref).abs().max()) < 6e-5        # logits up to +-4 at 16 mantissa bits per operand
    hip.softmax_rows_(s, mk)
    p_ref = torch.softmax(ref, dim=-1)
    assert float((s[:, :, :mk].double() - p_ref).abs().max()) < 2e-5
    o = torch.empty(n, 256, device=DEV)
    hip.gemm(s[:, :, :mk], vh.transpose(1, 2), o.view(n, 4, 64).permute(1, 0, 2))
    o_ref = torch.einsum("hnm,hmd->hnd", p_ref, vh.double()).permute(1, 0, 2).reshape(n, 256)
    assert float((o.double() - o_ref).abs().max()) < 3e-5
    # softmax backward
    dp = _rand(4, n, ld, seed=7)
    p_mine = s[:, :, :mk].double()
    dref = p_mine * (dp[:, :, :mk].double() - (dp[:, :, :mk].double() * p_mine).sum(-1, keepdim=True))
    hip.softmax_rows_backward_(s, dp, mk)
    assert float((dp[:, :, :mk].double() - dref).abs().max()) < 1e-6


@pytest.mark.parametrize("rows,c,relu", [((700, 300), 64, True), ((2048, 2031), 512, True), ((37,), 32, False), ((256, 257), 128, True)])
def test_batchnorm_train_forward_backward(rows, c, relu):
    """Per-segment batch statistics, running-statistics sequence, and the backward pass, against torch.nn.functional.batch_norm
    called once per segment (the reference calls the module once per image side)."""
    tot = sum(rows)
    x = _rand(tot, c, seed=1, scale=2.0) + 0.3
    gamma, beta = _rand(c, seed=2).abs() + 0.5, _rand(c, seed=3) * 0.1
    dy = _rand(tot, c, seed=4)
    rm, rv = _rand(c, seed=5) * 0.1, _rand(c, seed=6).abs() + 0.5
    offs = np.concatenate([[0], np.cumsum(rows)])
    sg = hip.segments([(int(offs[i]), int(rows[i])) for i in range(len(rows))])
    rm_h, rv_h = rm.clone(), rv.clone()
    y, save = hip.batchnorm_train_forward(x, sg, gamma, beta, 1e-5, 0.1, rm_h, rv_h, relu)
    dx, dg, db = hip.batchnorm_train_backward(x, dy, sg, save, gamma, beta, relu)
    with torch.enable_grad():           # (some test modules switch autograd off process-wide)
        xr = x.double().cpu().requires_grad_(True)
        gr, br = gamma.double().cpu().requires_grad_(True), beta.double().cpu().requires_grad_(True)
        rm_r, rv_r = rm.double().cpu(), rv.double().cpu()
        ys = []
        for i in range(len(rows)):
            seg = xr[offs[i]:offs[i + 1]].t()[None]                   # (1, C, N) like the reference's Conv1d activations
            o = torch.nn.functional.batch_norm(seg, rm_r, rv_r, gr, br, training=True, momentum=0.1, eps=1e-5)
            ys.append((o.relu() if relu else o)[0].t())
        yr = torch.cat(ys)
        yr.backward(dy.double().cpu())
    assert float((y.double().cpu() - yr.detach()).abs().max()) < 2e-5
    assert float((rm_h.double().cpu() - rm_r).abs().max()) < 1e-6 and float((rv_h.double().cpu() - rv_r).abs().max()) < 1e-5
    # entries whose pre-activation is within rounding of 0 may take the other branch of the ReLU: compare away from them
    sc = float(xr.grad.abs().max())
    assert float((dx.double().cpu() - xr.grad).abs().max()) < 2e-4 * sc
    assert float((dg.double().cpu() - gr.grad).abs().max()) < 2e-4 * float(gr.grad.abs().max())
    assert float((db.double().cpu() - br.grad).abs().max()) < 2e-4 * float(br.grad.abs().max())


def test_colsum_elementwise_permute():
    x = _rand(2031, 512, seed=1)
    s = hip.colsum(x)
    assert float((s.double() - x.double().sum(0)).abs().max()) < 2e-4
    s2 = hip.colsum(x[:300, :96], out=s[:96].clone(), beta=1.0)
    assert float((s2.double() - (x.double().sum(0)[:96] + x[:300, :96].double().sum(0))).abs().max()) < 2e-4
    assert torch.equal(hip.colsum(x), s)                              # deterministic
    a, b = _rand(100, 48, seed=2), _rand(100, 48, seed=3)
    out = torch.empty_like(a)
    assert torch.equal(hip.elementwise(hip.EW_ADD, out, a, b, alpha=1.0), a + b)
    assert torch.equal(hip.elementwise(hip.EW_RELU_MASK, out, a, b), torch.where(b > 0, a, torch.zeros_like(a)))
    # head interleave: weight rows d*4+h -> h*64+d
    w = _rand(256, 256, seed=4)
    wp = torch.empty_like(w)
    hip.permute3(wp, w, (64, 4, 256), (256, 64 * 256, 1), (4 * 256, 256, 1))
    assert torch.equal(wp, w.view(64, 4, 256).permute(1, 0, 2).reshape(256, 256))


def test_sage_mean_transposed_is_the_adjoint():
    """<mean(h), g> == <h, mean^T(g)> on a random symmetric graph."""
    n, c = 500, 128
    rng = np.random.default_rng(0)
    e = rng.integers(0, n, size=(2000, 2))
    e = e[e[:, 0] != e[:, 1]]
    adj = [set() for _ in range(n)]
    for u, v in e:
        adj[u].add(int(v)); adj[v].add(int(u))
    indptr = np.concatenate([[0], np.cumsum([len(a) for a in adj])]).astype(np.int32)
    indices = np.concatenate([sorted(a) for a in adj if a]).astype(np.int32)
    ip, ix = torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV)
    h, g = _rand(n, c, seed=1), _rand(n, c, seed=2)
    mean = torch.empty_like(h)
    hip.sage_mean(h, ip, ix, mean)
    gt = hip.sage_mean_transposed(g, ip, ix)
    lhs, rhs = float((mean.double() * g.double()).sum()), float((h.double() * gt.double()).sum())
    assert abs(lhs - rhs) < 1e-6 * max(1.0, abs(lhs))


@pytest.mark.parametrize("rows,c", [(300, 32), (1031, 512), (5, 256)])
def test_layernorm_backward(rows, c):
    """Reverse pass of the reference's LayerNorm (gmatcher.py:74-85: unbiased std over the channels, eps added to the std) + ReLU
    against torch autograd of that formula in float64."""
    x = _rand(rows, c, seed=1, scale=1.5) + 0.2
    a2, b2 = _rand(c, seed=2).abs() + 0.5, _rand(c, seed=3) * 0.1
    dy = _rand(rows, c, seed=4)
    y = torch.empty_like(x)
    hip.layernorm_act(x, a2, b2, out=y, act=hip.ACT_RELU)
    dx, da, db = hip.layernorm_backward(x, dy, a2, b2, True)
    with torch.enable_grad():
        xr = x.double().cpu().requires_grad_(True)
        ar, br = a2.double().cpu().requires_grad_(True), b2.double().cpu().requires_grad_(True)
        mean, std = xr.mean(-1, keepdim=True), xr.std(-1, keepdim=True)
        yr = (ar * (xr - mean) / (std + 1e-6) + br).relu()
        yr.backward(dy.double().cpu())
    assert float((y.double().cpu() - yr.detach()).abs().max()) < 1e-5
    for mine, ref in ((dx, xr.grad), (da, ar.grad), (db, br.grad)):
        assert float((mine.double().cpu() - ref).abs().max()) < 2e-4 * float(ref.abs().max())


def test_head_pack_roundtrip():
    """gims_head_pack: reference layout (channel = d * heads + h, gmatcher.py:108-113) -> head-contiguous rows / columns and back."""
    D, H = 256, 4
    pw = [_rand(D, D, 1, seed=10 + j) for j in range(3)]           # Conv1d weights [out, in, 1]
    pb = [_rand(D, seed=20 + j) for j in range(3)]
    mw = _rand(D, D, 1, seed=30)
    wqkv, bqkv, wm = torch.empty(3 * D, D, device=DEV), torch.empty(3 * D, device=DEV), torch.empty(D, D, device=DEV)
    hip.head_pack(pw, pb, mw, wqkv, bqkv, wm, H, to_params=False)
    for j in range(3):
        ref = pw[j].view(D // H, H, D).permute(1, 0, 2).reshape(D, D)
        assert torch.equal(wqkv[j * D:(j + 1) * D], ref)
        assert torch.equal(bqkv[j * D:(j + 1) * D], pb[j].view(D // H, H).t().reshape(D))
    assert torch.equal(wm, mw.view(D, D // H, H).permute(0, 2, 1).reshape(D, D))
    gw, gb, gm = [torch.empty_like(w) for w in pw], [torch.empty_like(b) for b in pb], torch.empty_like(mw)
    hip.head_pack(gw, gb, gm, wqkv, bqkv, wm, H, to_params=True)
    assert all(torch.equal(a, b) for a, b in zip(gw, pw)) and all(torch.equal(a, b) for a, b in zip(gb, pb)) and torch.equal(gm, mw)


def _attention_reference(qkv, do, problems, heads):
    """float64 torch: o, lse and the gradient of sum(o * do) with respect to qkv, problem by problem and head by head (gmatcher.py:35-39)."""
    rows, d3 = qkv.shape
    d, dh = d3 // 3, d3 // 3 // heads
    with torch.enable_grad():
        x = qkv.double().requires_grad_(True)
        o = torch.zeros((rows, d), dtype=torch.float64, device=qkv.device)
        lse = torch.zeros((heads, rows), dtype=torch.float64, device=qkv.device)
        outs = []
        for qo, nq, ko, nk in problems:
            q = x[qo:qo + nq, 0:d].view(nq, heads, dh).permute(1, 0, 2)
            k = x[ko:ko + nk, d:2 * d].view(nk, heads, dh).permute(1, 0, 2)
            v = x[ko:ko + nk, 2 * d:].view(nk, heads, dh).permute(1, 0, 2)
            s = q @ k.transpose(1, 2) / dh ** 0.5
            oo = (torch.softmax(s, -1) @ v).permute(1, 0, 2).reshape(nq, d)
            outs.append((qo, nq, oo))
            lse[:, qo:qo + nq] = torch.logsumexp(s, -1).detach()
        loss = sum((oo * do[qo:qo + nq].double()).sum() for qo, nq, oo in outs)
        loss.backward()
        for qo, nq, oo in outs:
            o[qo:qo + nq] = oo.detach()
    return o, lse, x.grad


@pytest.mark.parametrize("reverse", ["f32", "bf16x3"])
@pytest.mark.parametrize("splits", [None, 1, 3, 8])
@pytest.mark.parametrize("n0,n1,cross", [(300, 517, False), (300, 517, True), (33, 64, True), (1, 5, True), (128, 128, False), (2048, 1900, True)])
def test_train_attention_forward_backward_vs_float64(n0, n1, cross, splits, reverse, monkeypatch):
    """gims_train_attention_forward / _backward (flash-style, exact-f32 MFMA, no stored probabilities) against float64 autograd: self and cross
    problems of two images of different sizes, ragged tiles (sizes that are no multiples of 32 or 128), one-row images, peaked and diffuse rows in
    the same matrix; every split count of the streamed dimension (None = the library's own choice); the reverse pass in exact f32 products and in
    three bf16 passes (16-bit-mantissa products: 2e-4 of the largest gradient entry)."""
    prec = hip.PREC_BF16X3 if reverse == "bf16x3" else hip.PREC_F32
    if splits is not None:
        monkeypatch.setenv("GIMS_TRAIN_ATTN_SPLITS", str(splits))
    heads, d = 4, 256
    rows = n0 + n1
    qkv = _rand(rows, 3 * d, seed=n0 + n1, scale=1.0)
    qkv[:, :d] *= torch.linspace(0.3, 6.0, rows, device=DEV)[:, None]          # query norms from diffuse to sharply peaked rows
    do = _rand(rows, d, seed=7)
    problems = [(0, n0, n0, n1), (n0, n1, 0, n0)] if cross else [(0, n0, 0, n0), (n0, n1, n0, n1)]
    o, lse = hip.train_attention_forward(qkv, problems, heads)
    dqkv = hip.train_attention_backward(qkv, o, lse, do, problems, heads, precision=prec)
    ro, rlse, rg = _attention_reference(qkv, do, problems, heads)
    assert float((o.double() - ro).abs().max()) < 3e-6 * float(ro.abs().max())
    assert float((lse.double() - rlse).abs().max()) < 2e-6 + 4e-7 * float(rlse.abs().max())          # (a few ulp of the largest score)
    for j, name in enumerate("qkv"):
        g, r = dqkv[:, j * d:(j + 1) * d].double(), rg[:, j * d:(j + 1) * d]
        # (dS = P (dP - D) cancels completely where a row has one dominant source: the f32 rounding of dP and D, ~1e-6 of |dP|, is what is left)
        assert float((g - r).abs().max()) < (3e-5 if reverse == "f32" else 6e-4 if min(n0, n1) < 8 else 2e-4) * float(r.abs().max()), (name, float((g - r).abs().max()) / float(r.abs().max()))
    # deterministic: the same bits again
    o2, lse2 = hip.train_attention_forward(qkv, problems, heads)
    assert torch.equal(o, o2) and torch.equal(lse, lse2)
    assert torch.equal(dqkv, hip.train_attention_backward(qkv, o, lse, do, problems, heads, precision=prec))


@pytest.mark.parametrize("gain", [6.0, 24.0, 48.0])
def test_train_attention_reverse_precision_at_large_logits(gain):
    """the default reverse pass ('bf16x3') recomputes S from split-bf16 operand pairs (16 mantissa bits each) while lse comes from the
    exact-f32 forward, so P = exp(S' - lse) carries a relative error of about 2^-16 sum |q_i k_i|, which grows with the logit magnitude -- harmless
    on the fixtures (queries scaled up to 6 x), not bounded by them.  This test walks the magnitude up (|S| to about 25, 100 and 200 natural units)
    and pins what the two reverse precisions deliver against float64 autograd: exact-f32 products stay at 3e-5 of the largest gradient entry
    whatever the magnitude (measured 1.5e-6 / 1.3e-5 / 1.8e-5); the three-pass products deliver about 2.2e-6 |S| (5.8e-5 / 3.0e-4 / 5.8e-4: bounds
    below = twice the measured values) -- LINEAR in the logit magnitude, as the operand precision predicts.  train_backward_precision='f32' is the setting for sharply peaked trained attention."""
    heads, d, n0, n1 = 4, 256, 300, 517
    rows = n0 + n1
    qkv = _rand(rows, 3 * d, seed=4242, scale=1.0)
    qkv[:, :d] *= gain
    do = _rand(rows, d, seed=7)
    problems = [(0, n0, n0, n1), (n0, n1, 0, n0)]
    o, lse = hip.train_attention_forward(qkv, problems, heads)
    ro, rlse, rg = _attention_reference(qkv, do, problems, heads)
    s_mag = float(rlse.abs().max())
    # (the forward is exact-f32 arithmetic: its error against float64 is the f32 rounding of the scores themselves, a few ulp of |S| in the exponent)
    assert float((o.double() - ro).abs().max()) < max(3e-6, 2.5e-7 * s_mag) * float(ro.abs().max())
    errs = {}
    for prec, label in ((hip.PREC_F32, "f32"), (hip.PREC_BF16X3, "bf16x3")):
        dqkv = hip.train_attention_backward(qkv, o, lse, do, problems, heads, precision=prec)
        assert torch.isfinite(dqkv).all()
        errs[label] = float((dqkv.double() - rg).abs().max()) / float(rg.abs().max())
    print(f"gain {gain}: largest |lse| {s_mag:.1f}; reverse-pass error / largest gradient entry: f32 {errs['f32']:.2e}, bf16x3 {errs['bf16x3']:.2e}")
    assert errs["f32"] < 3e-5
    assert errs["bf16x3"] < BF16X3_REVERSE_BOUND[gain]


# measured on MI355X (round 6): 5.8e-5 / 3.0e-4 / 5.8e-4 at |S| = 33 / 134 / 268 -- about 2.2e-6 |S|; bounds = twice that
BF16X3_REVERSE_BOUND = {6.0: 1.2e-4, 24.0: 6e-4, 48.0: 1.2e-3}


def test_train_attention_more_problems_than_one_launch_takes():
    """34 problems (17 pairs of small images, cross): the problem table travels in the kernel arguments 32 at a time, so this call is two rounds
    of launches sharing one workspace."""
    heads, d = 4, 256
    sizes = [37 + 3 * i for i in range(34)]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    rows = int(offs[-1])
    qkv = _rand(rows, 3 * d, seed=99)
    do = _rand(rows, d, seed=98)
    problems = []
    for b in range(17):
        i0, i1 = 2 * b, 2 * b + 1
        problems += [(int(offs[i0]), sizes[i0], int(offs[i1]), sizes[i1]), (int(offs[i1]), sizes[i1], int(offs[i0]), sizes[i0])]
    o, lse = hip.train_attention_forward(qkv, problems, heads)
    ro, rlse, rg = _attention_reference(qkv, do, problems, heads)
    assert float((o.double() - ro).abs().max()) < 3e-6 * float(ro.abs().max())
    for prec, tol in ((hip.PREC_F32, 3e-5), (hip.PREC_BF16X3, 2e-4)):
        dqkv = hip.train_attention_backward(qkv, o, lse, do, problems, heads, precision=prec)
        assert float((dqkv.double() - rg).abs().max()) < tol * float(rg.abs().max())


def test_train_attention_rejects_bad_arguments():
    qkv = _rand(64, 768, seed=1)
    with pytest.raises(hip.GimsHipError):
        hip.train_attention_forward(qkv, [(0, 64, 0, 65)], 4)                 # sources beyond the rows
    with pytest.raises(hip.GimsHipError):
        hip.train_attention_forward(qkv, [(0, 0, 0, 64)], 4)                  # no queries
    with pytest.raises(hip.GimsHipError):
        hip.train_attention_forward(qkv[:, :384].contiguous(), [(0, 64, 0, 64)], 4)      # head dimension 32

"""One training step of GMatcher on the HIP path (SURVEY row f3 = a25): the forward pass of ``forward_train``
(models/gmatcher.py:309-386) with the module in train() mode -- BatchNorm on batch statistics, running statistics updated --
and the reverse pass through every stage of it, so that ``loss.backward()`` of train.py:136-137 fills ``.grad`` of all 282
parameters (``use_layernorm=True``: the reference's LayerNorm instead of BatchNorm in every MLP, same path).  Host orchestration only: every product runs in gims_gemm_f32 (split-bf16 MFMA, f32 class by default), the norms, softmaxes, sums
and graph aggregations in the kernels of csrc/train.hip, the Sinkhorn solve and its reverse sweep in csrc/sinkhorn.hip.

Layout: rows of all images SIDE-major ([image 0 of every batch element | image 1 of every batch element]), activations
row-major f32 [rows][channels].  The linear layers of a GNN layer run once over all rows (both sides share the weights,
gmatcher.py:139-141); BatchNorm statistics are per side (= per call of the module in the reference); attention runs per
image WITHOUT stored probabilities (one call per layer for all images and heads, csrc/train_attn.hip: the forward keeps the row
statistic max + log(sum), the reverse pass recomputes P from it; rounds 1-4 kept P, 2.4 GB at 2 x 2048 keypoints, and ran seven
products and two softmax passes per image and layer over it).  Heads are made contiguous by permuting the projection weights on the way in
(reference: channel = d * heads + h, gmatcher.py:108-113) and the weight gradients on the way out (gims_head_pack: one launch per
layer and direction).
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from . import hip

HEADS = 4
BN_EPS, BN_MOMENTUM = 1e-5, 0.1        # nn.BatchNorm1d defaults (gmatcher.py:20)


def _w2(p):
    """Conv1d(k=1) weight [out, in, 1] (or Linear [out, in]) as a 2-D matrix."""
    return p.detach().view(p.shape[0], -1)


def _precision(cfg):
    """config['train_precision']: 'bf16x6' (default: three bf16 parts per f32 operand, six MFMA passes -- the accuracy class of the
    reference's f32 matmuls, which the reverse pass needs: bias-like gradients are sums with heavy cancellation and a ReLU
    of a near-zero pre-activation falls the other way ~100x more often at 16 bits) or 'bf16x3' (two parts, three passes)."""
    p = cfg.get('train_precision', 'bf16x6')
    if p not in ('bf16x6', 'bf16x3'):
        raise ValueError("train_precision must be 'bf16x6' or 'bf16x3'")
    return hip.PREC_BF16X6 if p == 'bf16x6' else hip.PREC_BF16X3


def _backward_precision(cfg):
    """config['train_backward_precision']: precision of the products of the REVERSE pass, default 'bf16x3' whatever the forward runs in.  The
    reverse pass is linear in its operands -- no ReLU decides anything there -- and measured on every trainstep_* fixture the gradient errors
    with a 3-pass reverse pass equal those of a 6-pass one to three digits (worst 1.0e-3 ... 1.7e-2, p95 8e-5 ... 4.7e-4 either way: what is
    left is the forward's ReLU flips)."""
    p = cfg.get('train_backward_precision', 'bf16x3')
    if p not in ('bf16x6', 'bf16x3'):
        raise ValueError("train_backward_precision must be 'bf16x6' or 'bf16x3'")
    return hip.PREC_BF16X6 if p == 'bf16x6' else hip.PREC_BF16X3


class _Step:
    """Everything the reverse pass needs from the forward pass of one step."""


def _params(model):
    """(names, parameters) of the module, cached on it (walking the module tree three times per step cost ~2 ms of host time);
    invalidated with the pack cache when parameters are replaced (GMatcher._apply / load_state_dict)."""
    c = model.__dict__.get("_train_params")
    if c is None or model.__dict__.get("_plist") is None:
        named = list(model.named_parameters())
        c = model.__dict__["_train_params"] = (tuple(n for n, _ in named), [p for _, p in named])
        model.__dict__.setdefault("_plist", c[1])
    return c


def _sage_bias(P, i):
    k = f"gnn_encoder.layers.{i}.fc_self.bias"
    return k if k in P else f"gnn_encoder.layers.{i}.bias"


# ------------------------------------------------------------------------------------------------ forward
def forward(model, data):
    """Returns (out3 = [loss, pos_loss, neg_loss] device tensor, _Step).  Mutates ``data`` like the reference's forward does
    (gmatcher.py:244-252) and updates the BatchNorm buffers of ``model``."""
    with hip.gemm_precision(_precision(model.config)):
        out = _forward(model, data)
    # the running statistics just changed through raw pointers (no tensor version bump): an inference pack with the old
    # statistics folded in must not survive a train() -> eval() switch
    model._pack = None
    return out


def _forward(model, data):
    cfg = model.config
    ln = bool(cfg['use_layernorm'])          # MLP(): Conv1d -> LayerNorm -> ReLU instead of Conv1d -> BatchNorm1d -> ReLU (gmatcher.py:17-23)
    radius, percentile, min_size = data.get('radius', 25), data.get('percentile', 7), data.get('min_size', 8)
    B = data['keypoints0'].shape[0]
    D = cfg['descriptor_dim']
    # side-major image order: [b0 s0, b1 s0, ..., b0 s1, b1 s1, ...]
    images = model._ingest([(data['keypoints' + side][b], data['descriptors' + side][b], data['scores' + side][b], data['image' + side].shape)
                            for side in ("0", "1") for b in range(B)])
    G, params = None, (radius, percentile, min_size)
    while G is None:
        ctx = model._run_build(images, *params)
        G = model._gather(ctx)
        params = ctx["params"]                      # (a build that has to be repeated says how: larger capacity, robust percentile flow)
    model._finish_graphs(images, G)
    dev = G["feat"].device
    for s in range(2):
        if len({images[s * B + b]["n_kept"] for b in range(B)}) != 1:
            raise RuntimeError("stack expects each tensor to be equal size: the batch elements keep different numbers of keypoints "
                               "(torch.stack in models/gmatcher.py:244-249 needs equal counts)")
    for s, side in enumerate(("0", "1")):
        gs = [images[s * B + b]["graph"] for b in range(B)]
        data['keypoints' + side] = torch.stack([h.ndata['point'] for h in gs])
        data['descriptors' + side] = torch.stack([h.ndata['feat'] for h in gs]).permute(0, 2, 1)
        data['scores' + side] = torch.stack([h.ndata['score'] for h in gs])
        data['kept_kpts%s_indices' % side] = [images[s * B + b]["kept"].tolist() for b in range(B)]
        data['graph' + side] = gs

    P = {k: v.detach() for k, v in zip(*_params(model))}
    Bf = model.__dict__.get("_train_buffers")        # (walking the module tree for the buffers cost 0.5 ms per step; dropped with _train_params)
    if Bf is None or model.__dict__.get("_train_params") is None:
        _params(model)
        Bf = model.__dict__["_train_buffers"] = dict(model.named_buffers())
    n_tot = G["n_tot"]
    rows = [[images[s * B + b]["rows"] for s in range(2)] for b in range(B)]          # rows[b][s] = (offset, n)
    side_rows = [(images[s * B]["rows"][0], sum(images[s * B + b]["n_kept"] for b in range(B))) for s in range(2)]
    sg = hip.segments(side_rows)
    S = _Step()
    S.images, S.G, S.rows, S.sg, S.B, S.n_tot, S.D, S.ln, S.P = images, G, rows, sg, B, n_tot, D, ln, P

    def bn(prefix, x, relu=True):
        if ln:                                # the reference's LayerNorm has no state: the reverse pass recomputes everything from x
            y = torch.empty_like(x)
            hip.layernorm_act(x, P[prefix + ".a_2"], P[prefix + ".b_2"], out=y, act=hip.ACT_RELU if relu else hip.ACT_NONE)
            return y, None
        y, save = hip.batchnorm_train_forward(x, sg, P[prefix + ".weight"], P[prefix + ".bias"], BN_EPS, BN_MOMENTUM,
                                              Bf.get(prefix + ".running_mean"), Bf.get(prefix + ".running_var"), relu)
        nbt = Bf.get(prefix + ".num_batches_tracked")
        if nbt is not None:
            nbt += sg.n                       # one forward call per side in the reference
        return y, save

    # ---- GraphSAGE (gmatcher.py:145-162; SAGEConv 'mean': fc_neigh before the aggregation iff in > out)
    h = G["feat"]
    S.sage = []
    for i in range(3):
        pre = f"gnn_encoder.layers.{i}."
        ws, wn, bs = P[pre + "fc_self.weight"], P[pre + "fc_neigh.weight"], P[_sage_bias(P, i)]
        act = hip.ACT_RELU if i < 2 else hip.ACT_NONE
        if ws.shape[1] > ws.shape[0]:
            xn = hip.gemm(h, wn)
            agg = torch.empty_like(xn)
            hip.sage_mean(xn, G["indptr_all"], G["indices_all"], agg)
            out = hip.gemm(h, ws, bias=bs, residual=agg, act=act)
            S.sage.append(dict(h=h, before=True, out=out))
        else:
            agg = torch.empty_like(h)
            hip.sage_mean(h, G["indptr_all"], G["indices_all"], agg)
            out = hip.gemm(h, ws, bias=bs)
            hip.gemm(agg, wn, out, beta=1.0, act=act)
            S.sage.append(dict(h=h, before=False, agg=agg, out=out))
        h = out
    sage = h

    # ---- keypoint encoder (gmatcher.py:26-33, 87-97): conv -> BN(train) -> ReLU ..., last conv + sage
    x = hip.normalize_keypoints(G["kpts_all"], G["norm3"], G["seg"])
    S.kenc = []
    n_convs = len(cfg['keypoint_encoder']) + 1
    idx = 0
    for i in range(n_convs):
        w, bia = _w2(P[f"kenc.encoder.{idx}.weight"]), P[f"kenc.encoder.{idx}.bias"]
        if i == n_convs - 1:
            # the residual stream of layer l lives in the LEFT half of that layer's [x | msg] buffer (the MLP of a layer reads
            # cat([x, msg]), gmatcher.py:123: one product over k = 2D instead of two, and one weight-gradient product)
            xm = torch.empty((n_tot, 2 * D), dtype=torch.float32, device=dev)
            y = hip.gemm(x, w, xm[:, :D], bias=bia, residual=sage)
            S.kenc.append(dict(x=x, conv=idx))
        else:
            pre = hip.gemm(x, w, bias=bia)
            y, save = bn(f"kenc.encoder.{idx + 1}", pre)
            S.kenc.append(dict(x=x, conv=idx, pre=pre, save=save))
            idx += 3
        x = y
    desc, xm_cur = x, xm

    # ---- attentional GNN (gmatcher.py:99-143)
    S.layers = []
    # problem tables of the two layer kinds: every image attends to itself ('self') or to the other image of its pair ('cross')
    S.attn_problems = {cross: hip.train_attn_problems([(*rows[b][s], *rows[b][1 - s if cross else s]) for b in range(B) for s in range(2)])
                       for cross in (False, True)}
    for l, name in enumerate(cfg['transformer_layers']):
        pre = f"gnn.layers.{l}."
        wqkv = torch.empty((3 * D, D), dtype=torch.float32, device=dev)
        bqkv = torch.empty(3 * D, dtype=torch.float32, device=dev)
        wm = torch.empty((D, D), dtype=torch.float32, device=dev)
        hip.head_pack([P[pre + f"attn.proj.{j}.weight"] for j in range(3)], [P[pre + f"attn.proj.{j}.bias"] for j in range(3)],
                      P[pre + "attn.merge.weight"], wqkv, bqkv, wm, HEADS, to_params=False)
        xm = xm_cur                                  # [x | msg] of this layer; desc is its left half
        qkv = hip.gemm(desc, wqkv, bias=bqkv)
        # attention of all images and heads in one call; the reverse pass recomputes the probabilities from lse (csrc/train_attn.hip)
        o, lse = hip.train_attention_forward(qkv, S.attn_problems[name == 'cross'], HEADS)
        msg = hip.gemm(o, wm, xm[:, D:], bias=P[pre + "attn.merge.bias"])
        w0, w3 = _w2(P[pre + "mlp.0.weight"]), _w2(P[pre + "mlp.3.weight"])
        hpre = hip.gemm(xm, w0, bias=P[pre + "mlp.0.bias"])
        hid, save = bn(pre + "mlp.1", hpre)
        xm_next = torch.empty((n_tot, 2 * D), dtype=torch.float32, device=dev)
        nxt = hip.gemm(hid, w3, xm_next[:, :D], bias=P[pre + "mlp.3.bias"], residual=desc)       # desc + delta (gmatcher.py:142)
        S.layers.append(dict(x=desc, xm=xm, wqkv=wqkv, wm=wm, qkv=qkv, o=o, lse=lse, msg=msg, hpre=hpre, save=save, hid=hid, cross=name == 'cross'))
        desc, xm_cur = nxt, xm_next
    S.desc = desc

    # ---- final projection, scores, Sinkhorn, loss (gmatcher.py:330-385)
    mdesc = hip.gemm(desc, _w2(P["final_proj.weight"]), bias=P["final_proj.bias"])
    S.mdesc = mdesc
    items = []
    for b in range(B):
        (o0, n0), (o1, n1) = rows[b]
        ld = (n1 + 3) // 4 * 4
        scores = torch.empty((n0, ld), dtype=torch.float32, device=dev)
        hip.gemm(mdesc[o0:o0 + n0], mdesc[o1:o1 + n1], scores[:, :n1], alpha=1.0 / math.sqrt(D))
        items.append(dict(scores=scores, n=n0, m=n1, matches0=torch.empty(n0, dtype=torch.int64, device=dev),
                          matches1=torch.empty(n1, dtype=torch.int64, device=dev), mscores0=torch.empty(n0, dtype=torch.float32, device=dev),
                          mscores1=torch.empty(n1, dtype=torch.float32, device=dev), uv=torch.empty(n0 + n1 + 3, dtype=torch.float32, device=dev)))
    S.items = items
    S.alpha = float(P["bin_score"])
    # ONE Sinkhorn solve serves both passes: the streamed kernels with the potentials after every iteration recorded (what the
    # reverse sweep needs); the loss reads the final potentials it leaves in `uv`
    S.hists = hip.sinkhorn_history(items, S.alpha, cfg['sinkhorn_iterations'])
    gt = data['matches'].to(device=dev, dtype=torch.int64).contiguous()
    out3, _ = hip.train_loss(items, [images[b]["kept"] for b in range(B)], [images[B + b]["kept"] for b in range(B)], gt, S.alpha,
                             cfg['pos_loss_weight'], cfg['neg_loss_weight'])
    S.loss_state = hip.train_loss.last
    return out3, S


# ------------------------------------------------------------------------------------------------ backward
_SIDE_STREAM = os.environ.get("GIMS_TRAIN_SIDE_STREAM", "1") != "0"      # A/B switch: 0 = parameter gradients on the main stream
_side_streams = {}


def _side_stream(dev):
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    s = _side_streams.get(key)
    if s is None:
        s = _side_streams[key] = torch.cuda.Stream(device=dev)
    return s


def backward(model, S, w_pos: float, w_neg: float):
    """Gradients of  w_pos * (pos_loss / pos_loss_weight) + w_neg * (neg_loss / neg_loss_weight)  -- i.e. with w_pos / w_neg the
    effective weights of the two loss terms -- with respect to every parameter: dict name -> tensor shaped like the parameter."""
    with hip.gemm_precision(_backward_precision(model.config)):
        return _backward(model, S, w_pos, w_neg)


def _backward(model, S, w_pos: float, w_neg: float):
    cfg = model.config
    P = S.P                                       # the detached parameter views of the forward pass (same storage)
    D, B, n_tot, rows, sg, G = S.D, S.B, S.n_tot, S.rows, S.sg, S.G
    dev = S.mdesc.device
    grads = {}

    def put(name, g):
        grads[name] = g.view(P[name].shape)

    # Parameter gradients (weight-gradient products with their split-K folds, bias column sums, the head un-permutation) are off the
    # critical path: nothing in the reverse pass reads them.  They go to a SIDE STREAM and fill the CUs that the narrow products of the
    # activation-gradient chain (64-192 tiles per launch) leave idle.  Every job waits for an event recorded on the main stream at its
    # submission (its inputs are complete then); its input tensors are held until the streams have joined at the end, and the main
    # stream never writes a buffer in place that a job may still be reading (dx and dqkv are fresh tensors per layer).
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev) if _SIDE_STREAM else None
    hold = []

    pending = []

    def aside(fn, *inputs):
        """Queue a parameter-gradient job.  Jobs are submitted in batches (flush): one event per batch instead of one per job -- record + wait
        cost 7 us of host time each, 110 of them per step, on a step that is bound by its host side."""
        if side is None:
            fn()
            return
        hold.extend(inputs)
        pending.append(fn)

    def flush():
        if not pending:
            return
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with hip.on_stream(side.cuda_stream):
            for fn in pending:
                fn()
        pending.clear()

    def norm_backward(prefix, x_pre, g, save):
        """Reverse pass of the norm + ReLU behind a convolution: returns the gradient of the convolution's output and stores
        the gradients of the norm's two parameter vectors."""
        if S.ln:
            dxn, dscale, dshift = hip.layernorm_backward(x_pre, g, P[prefix + ".a_2"], P[prefix + ".b_2"], True)
            put(prefix + ".a_2", dscale)
            put(prefix + ".b_2", dshift)
            return dxn
        dxn, dgam, dbet = hip.batchnorm_train_backward(x_pre, g, sg, save, P[prefix + ".weight"], P[prefix + ".bias"], True)
        put(prefix + ".weight", dgam)
        put(prefix + ".bias", dbet)
        return dxn

    # ---- loss -> scores (reverse sweep through the unrolled Sinkhorn iterations) -> matching descriptors
    dscores, dalpha = hip.sinkhorn_score_gradients(S.items, S.alpha, cfg['sinkhorn_iterations'], w_pos, w_neg, S.loss_state, S.hists)
    put("bin_score", dalpha)
    dm = torch.empty_like(S.mdesc)
    inv = 1.0 / math.sqrt(D)
    for b in range(B):
        (o0, n0), (o1, n1) = rows[b]
        hip.gemm(dscores[b], S.mdesc[o1:o1 + n1].t(), dm[o0:o0 + n0], alpha=inv)
        hip.gemm(dscores[b].t(), S.mdesc[o0:o0 + n0].t(), dm[o1:o1 + n1], alpha=inv)
    wf = _w2(P["final_proj.weight"])
    aside(lambda: (put("final_proj.weight", hip.gemm(dm.t(), S.desc.t())), put("final_proj.bias", hip.colsum(dm))), dm, S.desc)
    dx = hip.gemm(dm, wf.t())                                     # gradient w.r.t. the residual stream after the last layer

    # ---- GNN layers in reverse
    for l in range(len(S.layers) - 1, -1, -1):
        L = S.layers[l]
        pre = f"gnn.layers.{l}."
        w0, w3 = _w2(P[pre + "mlp.0.weight"]), _w2(P[pre + "mlp.3.weight"])
        # delta = mlp(cat[x, msg]); x_next = x + delta: dx is d/dx_next = d/ddelta
        aside(lambda dx=dx, L=L, pre=pre: (put(pre + "mlp.3.weight", hip.gemm(dx.t(), L["hid"].t())), put(pre + "mlp.3.bias", hip.colsum(dx))), dx, L["hid"])
        dhid = hip.gemm(dx, w3.t())
        dhpre = norm_backward(pre + "mlp.1", L["hpre"], dhid, L["save"])
        aside(lambda dhpre=dhpre, L=L, pre=pre: (put(pre + "mlp.0.weight", hip.gemm(dhpre.t(), L["xm"].t())), put(pre + "mlp.0.bias", hip.colsum(dhpre))), dhpre, L["xm"])
        dx = hip.gemm(dhpre, w0[:, :D].t(), residual=dx)          # dx + dhpre W0[:, :D]   (x enters the MLP directly); a fresh tensor: see aside
        dmsg = hip.gemm(dhpre, w0[:, D:].t())
        # merge
        aside(lambda dmsg=dmsg, pre=pre: put(pre + "attn.merge.bias", hip.colsum(dmsg)), dmsg)
        do = hip.gemm(dmsg, L["wm"].t())
        # attention of all images and heads (every image's rows are queries once and sources once per layer: dqkv is written exactly once)
        dqkv = hip.train_attention_backward(L["qkv"], L["o"], L["lse"], do, S.attn_problems[L["cross"]], HEADS)
        gw = [torch.empty_like(P[pre + f"attn.proj.{j}.weight"]) for j in range(3)]
        gb = [torch.empty_like(P[pre + f"attn.proj.{j}.bias"]) for j in range(3)]
        gm = torch.empty_like(P[pre + "attn.merge.weight"])

        def qkv_grads(dqkv=dqkv, dmsg=dmsg, L=L, gw=gw, gb=gb, gm=gm):
            dwm = hip.gemm(dmsg.t(), L["o"].t())      # in the packed (head-contiguous) layout; unpacked with the projections
            dwqkv = hip.gemm(dqkv.t(), L["x"].t())
            dbqkv = hip.colsum(dqkv)
            hip.head_pack(gw, gb, gm, dwqkv, dbqkv, dwm, HEADS, to_params=True)
            hold.extend((dwm, dwqkv, dbqkv))
        aside(qkv_grads, dqkv, L["x"], dmsg, L["o"])
        for j in range(3):
            put(pre + f"attn.proj.{j}.weight", gw[j])
            put(pre + f"attn.proj.{j}.bias", gb[j])
        put(pre + "attn.merge.weight", gm)
        dx = hip.gemm(dqkv, L["wqkv"].t(), residual=dx)           # dx + dQKV Wqkv
        flush()
        S.layers[l] = None                                        # this layer's activations are no longer needed (the queued jobs hold what they read)

    # ---- keypoint encoder (dx is now d/d(sage + kenc))
    ddesc = dx
    g = ddesc
    for i in range(len(S.kenc) - 1, -1, -1):
        K = S.kenc[i]
        idx = K["conv"]
        if "pre" in K:           # conv idx -> BN idx+1 -> ReLU: g is the gradient of the ReLU output
            g = norm_backward(f"kenc.encoder.{idx + 1}", K["pre"], g, K["save"])
        aside(lambda g=g, K=K, idx=idx: (put(f"kenc.encoder.{idx}.weight", hip.gemm(g.t(), K["x"].t())), put(f"kenc.encoder.{idx}.bias", hip.colsum(g))), g, K["x"])
        if i > 0:
            g = hip.gemm(g, _w2(P[f"kenc.encoder.{idx}.weight"]).t())

    # ---- GraphSAGE
    g = ddesc
    for i in range(2, -1, -1):
        L = S.sage[i]
        pre = f"gnn_encoder.layers.{i}."
        ws, wn = P[pre + "fc_self.weight"], P[pre + "fc_neigh.weight"]
        if i < 2:                 # ReLU after layers 0 and 1
            g = hip.elementwise(hip.EW_RELU_MASK, torch.empty_like(g), g, L["out"])
        aside(lambda g=g, L=L, pre=pre, i=i: (put(pre + "fc_self.weight", hip.gemm(g.t(), L["h"].t())), put(_sage_bias(P, i), hip.colsum(g))), g, L["h"])
        if L["before"]:           # out = h Ws^T + b + mean(h Wn^T)
            dxn = hip.sage_mean_transposed(g, G["indptr_all"], G["indices_all"])
            aside(lambda dxn=dxn, L=L, pre=pre: put(pre + "fc_neigh.weight", hip.gemm(dxn.t(), L["h"].t())), dxn, L["h"])
            if i > 0:
                gh = hip.gemm(g, ws.t())
                hip.gemm(dxn, wn.t(), gh, beta=1.0)
                g = gh
        else:                     # out = h Ws^T + b + mean(h) Wn^T
            aside(lambda g=g, L=L, pre=pre: put(pre + "fc_neigh.weight", hip.gemm(g.t(), L["agg"].t())), g, L["agg"])
            if i > 0:
                dagg = hip.gemm(g, wn.t())
                gh = hip.gemm(g, ws.t())
                hip.elementwise(hip.EW_ACC, gh, hip.sage_mean_transposed(dagg, G["indptr_all"], G["indices_all"]), alpha=1.0)
                g = gh
    flush()
    if side is not None:                                          # join: the caller (and the allocator's reuse of everything held) comes after the side jobs
        ev = torch.cuda.Event()
        ev.record(side)
        main.wait_event(ev)
    return grads


# ------------------------------------------------------------------------------------------------ autograd boundary
class _TrainStepFn(torch.autograd.Function):
    """(loss, pos_loss, neg_loss) as differentiable functions of the module's parameters: forward and backward are the HIP
    passes above; the parameters are the inputs so that .grad, optimizers, DDP hooks and GradScaler see an ordinary graph."""

    @staticmethod
    def forward(ctx, model, data, names, *params):
        with hip.pinned_stream():
            out3, S = forward(model, data)
        ctx.model, ctx.S, ctx.names = model, S, names
        return out3[0].clone(), out3[1].clone(), out3[2].clone()

    @staticmethod
    def backward(ctx, g_loss, g_pos, g_neg):
        model, S = ctx.model, ctx.S
        if S is None:
            raise RuntimeError("the activations of this training step were already released (backward called twice)")
        zero = torch.zeros((), dtype=torch.float32, device=S.mdesc.device)
        gl, gp, gn = torch.stack([zero if g is None else g.float().reshape(()) for g in (g_loss, g_pos, g_neg)]).tolist()
        cfg = model.config
        # loss = pos_loss + neg_loss, pos_loss = pos_weight * mean-of-means, neg_loss likewise (gmatcher.py:383-385)
        with hip.pinned_stream():
            grads = backward(model, S, cfg['pos_loss_weight'] * (gl + gp), cfg['neg_loss_weight'] * (gl + gn))
        ctx.S = None
        return (None, None, None) + tuple(grads.get(n) for n in ctx.names)


def train_forward(model, data):
    """``GMatcher.forward(data, mode='train')`` for a module in train() mode: returns (loss, pos_loss, neg_loss), 0-dim tensors
    attached to the autograd graph of the module's parameters."""
    names, params = _params(model)
    return _TrainStepFn.apply(model, data, names, *params)

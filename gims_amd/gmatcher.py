"""Drop-in ``GMatcher`` for the GIMS matcher hot path, running on MI355X through libgims_hip.so.

Mirrors the reference's operator interface (models/gmatcher.py):
  * ``GMatcher(config)`` with the same ``default_config`` keys (gmatcher.py:166-176) and checkpoint
    handling (``ema`` -> ``model`` -> raw state dict, gmatcher.py:208-217);
  * ``state_dict()`` / ``load_state_dict()`` use the reference's 348 parameter names (``bin_score``,
    ``kenc.encoder.*``, ``gnn.layers.*.attn.{merge,proj.N}``, ``gnn.layers.*.mlp.*``,
    ``gnn_encoder.layers.*.{fc_self,fc_neigh}``, ``final_proj``), both SAGEConv bias layouts accepted;
  * ``forward(data)`` consumes and MUTATES the same dict (gmatcher.py:219-307): kept keypoints /
    descriptors / scores, ``kept_kpts{0,1}_indices``, ``graph0/1``; returns the same result dict
    (int64 ``matches0/1`` with -1 for no match, f32 ``matching_scores0/1``, ``mdesc0/1`` ...).

Host code is Python on PyTorch-ROCm (device memory + streams); all arithmetic of the path runs in the
HIP kernels of ``gims_amd/csrc`` through the C ABI of ``include/gims_hip.h``.  No CPU fallback exists.
Internally activations are point-major ([rows, channels]); all images of a call are concatenated row-wise
so that every linear layer is ONE launch for the whole batch.
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch
import torch.nn as nn

from . import hip

BN_EPS = 1e-5


class _Node(nn.Module):
    """Bare container used to reproduce the reference's parameter tree (names only, no forward)."""


def _register(root: nn.Module, dotted: str, tensor: torch.Tensor, buffer: bool):
    mod = root
    parts = dotted.split(".")
    for p in parts[:-1]:
        if not hasattr(mod, p):
            mod.add_module(p, _Node())
        mod = getattr(mod, p)
    if buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


class GraphHandle:
    """What the reference hands back as ``data['graph0/1'][b]`` (a DGLGraph there): CSR of the adaptive
    graph over the kept keypoints (both edge directions), plus the node data the reference stores on it."""

    def __init__(self, indptr, indices, ndata):
        self.indptr, self.indices, self.ndata = indptr, indices, ndata

    def num_nodes(self):
        return int(self.indptr.numel() - 1)

    def num_edges(self):
        return int(self.indices.numel())

    def edges(self):
        deg = (self.indptr[1:] - self.indptr[:-1]).long()
        dst = torch.repeat_interleave(torch.arange(deg.numel(), device=deg.device), deg)
        return self.indices.long(), dst


class GMatcher(nn.Module):
    default_config = {
        'descriptor_dim': 256,
        'weights_path': None,
        'keypoint_encoder': [32, 64, 128, 256],
        'transformer_layers': ['self', 'cross'] * 9,
        'sinkhorn_iterations': 100,
        'match_threshold': 0.2,
        'use_layernorm': False,
        'input_dim': 256,
        'num_heads': 4,
        # --- additions (defaults keep the reference behaviour) ---
        'linear_precision': 'bf16x3',   # 'bf16x3' (split-bf16 MFMA, ~2^-17) or 'f32' (exact-f32 MFMA)
        'verbose': False,               # the reference prints '>> ...' timing lines; off by default here
    }

    def __init__(self, config):
        super().__init__()
        self.config = {**self.default_config, **config}
        cfg = self.config
        if cfg['use_layernorm']:
            raise NotImplementedError("use_layernorm=True (gmatcher.py:74-85) is not on the HIP path yet")
        if cfg['input_dim'] != cfg['descriptor_dim']:
            raise NotImplementedError("input_proj is built but never called by the reference forward (gmatcher.py:198-201)")
        D = cfg['descriptor_dim']
        if D != 256:
            raise NotImplementedError("descriptor_dim must be 256 (4 heads x 64)")
        self.n_layers = len(cfg['transformer_layers'])
        self._heads = 4   # AttentionalGNN hard-codes 4 heads (gmatcher.py:131); config['num_heads'] is ignored there too
        from .synth import state_dict_spec
        for name, shape in state_dict_spec(D, tuple(cfg['keypoint_encoder']), self.n_layers):
            if name.endswith("num_batches_tracked"):
                _register(self, name, torch.zeros((), dtype=torch.int64), buffer=True)
            elif name.endswith("running_mean"):
                _register(self, name, torch.zeros(shape), buffer=True)
            elif name.endswith("running_var"):
                _register(self, name, torch.ones(shape), buffer=True)
            elif name == "bin_score":
                _register(self, name, torch.tensor(1.0), buffer=False)
            else:
                _register(self, name, torch.zeros(shape), buffer=False)
        self._pack = None
        self._pack_key = None
        if cfg['weights_path']:
            weights = torch.load(cfg['weights_path'], map_location="cpu", weights_only=False)
            if ('ema' in weights) and (weights['ema'] is not None):
                load_dict = weights['ema']
            elif 'model' in weights:
                load_dict = weights['model']
            else:
                load_dict = weights
            self.load_state_dict(load_dict)
            print('Loaded GMatcher model ("{}" weights)'.format(cfg['weights_path']))

    # ------------------------------------------------------------------ checkpoint compatibility
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        sd = {}
        for k, v in state_dict.items():
            k = k[7:] if k.startswith("module.") else k          # DDP prefix (utils/common.py:107-114)
            # older DGL: SAGEConv keeps a separate ``bias`` parameter instead of ``fc_self.bias``
            if k.startswith("gnn_encoder.layers.") and k.endswith(".bias") and k.count(".") == 3:
                k = k[:-len("bias")] + "fc_self.bias"
            sd[k] = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
        self._pack = None
        return super().load_state_dict(sd, strict=strict, **kw)

    # ------------------------------------------------------------------ weight packing
    def _packed(self, device):
        key = (str(device), self.config['linear_precision'], sum(int(p._version) for p in self.parameters()))
        if self._pack is not None and self._pack_key == key:
            return self._pack
        sd = {k: v.detach().to("cpu", torch.float32) for k, v in self.state_dict().items()}
        x3 = self.config['linear_precision'] == 'bf16x3'
        if self.config['linear_precision'] not in ('bf16x3', 'f32'):
            raise ValueError("linear_precision must be 'bf16x3' or 'f32'")
        P: Dict[str, object] = {"x3": x3}

        def fold(w, b, prefix):   # Conv1d(k=1) followed by BatchNorm1d(eval)  (gmatcher.py:17-22)
            g = sd[prefix + ".weight"] / torch.sqrt(sd[prefix + ".running_var"] + BN_EPS)
            return w * g[:, None], (b - sd[prefix + ".running_mean"]) * g + sd[prefix + ".bias"]

        def dev(t):
            return t.contiguous().to(device)

        def lin(w, b):   # a linear layer in the configured precision
            w = w.contiguous()
            e = {"b": dev(b), "n": w.shape[0], "k": w.shape[1]}
            if x3 and w.shape[1] % 64 == 0:
                hi, lo = hip.split_bf16(dev(w))
                e.update(w=hi, w_lo=lo, prec=hip.PREC_BF16X3)
            else:
                e.update(w=dev(w), w_lo=None, prec=hip.PREC_F32)
            return e

        # keypoint encoder: Sequential indices conv 0,3,6,9,12 / BN 1,4,7,10 (gmatcher.py:92)
        nk = len(self.config['keypoint_encoder']) + 1
        w, b = fold(sd["kenc.encoder.0.weight"][:, :, 0], sd["kenc.encoder.0.bias"], "kenc.encoder.1")
        P["kenc_w1"], P["kenc_b1"] = dev(w), dev(b)
        P["kenc"] = []
        for i in range(1, nk):
            w, b = sd[f"kenc.encoder.{3 * i}.weight"][:, :, 0], sd[f"kenc.encoder.{3 * i}.bias"]
            if i < nk - 1:
                w, b = fold(w, b, f"kenc.encoder.{3 * i + 1}")
            P["kenc"].append(lin(w, b))
        # GraphSAGE: [W_self | W_neigh] on [h | mean(h)]  (gmatcher.py:149-151)
        P["sage"] = []
        for i in range(3):
            p = f"gnn_encoder.layers.{i}."
            P["sage"].append(lin(torch.cat([sd[p + "fc_self.weight"], sd[p + "fc_neigh.weight"]], 1), sd[p + "fc_self.bias"]))
        # attentional GNN.  Heads are interleaved in the reference (channel c = d*H + h, gmatcher.py:111);
        # permute q/k/v output rows and merge input columns to head-blocked order c' = h*64 + d.
        H, D = self._heads, self.config['descriptor_dim']
        dh = D // H
        perm = torch.tensor([(c % dh) * H + (c // dh) for c in range(D)])   # new index c' -> old channel
        P["layers"] = []
        for l in range(self.n_layers):
            p = f"gnn.layers.{l}."
            wq, wk, wv = [sd[p + f"attn.proj.{j}.weight"][:, :, 0][perm] for j in range(3)]
            bq, bk, bv = [sd[p + f"attn.proj.{j}.bias"][perm] for j in range(3)]
            wm = sd[p + "attn.merge.weight"][:, :, 0][:, perm]
            w0, b0 = fold(sd[p + "mlp.0.weight"][:, :, 0], sd[p + "mlp.0.bias"], p + "mlp.1")
            P["layers"].append({
                "qkv": lin(torch.cat([wq, wk, wv], 0), torch.cat([bq, bk, bv], 0)),
                "merge": lin(wm, sd[p + "attn.merge.bias"]),
                "mlp0": lin(w0, b0),
                "mlp1": lin(sd[p + "mlp.3.weight"][:, :, 0], sd[p + "mlp.3.bias"]),
                "cross": self.config['transformer_layers'][l] == 'cross',
            })
        P["final"] = lin(sd["final_proj.weight"][:, :, 0], sd["final_proj.bias"])
        P["alpha"] = float(sd["bin_score"])
        self._pack, self._pack_key = P, key
        return P

    @staticmethod
    def _lin(e, a0, **kw):
        return hip.linear(a0, e["w"], w_lo=e["w_lo"], bias=e["b"], precision=e["prec"], **kw)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, data, **kwargs):
        cfg = self.config
        radius = data.get('radius', 25)
        percentile = data.get('percentile', 7)
        min_size = data.get('min_size', 8)
        if data.get('delaunay', False):
            raise NotImplementedError("delaunay=True is broken in the reference snapshot (UnboundLocalError, gmatcher.py:250)")
        if kwargs.get('mode', 'test') == "train":
            raise NotImplementedError("forward_train (gmatcher.py:309-386) is not on the HIP path yet")
        dev = data['keypoints0'].device
        if dev.type != "cuda":
            raise hip.GimsHipError("GMatcher runs on the GPU only (no CPU fallback): move the inputs to 'cuda'")
        P = self._packed(dev)
        B = data['keypoints0'].shape[0]
        D = cfg['descriptor_dim']

        # ---- adaptive graph construction, all images of the call enqueued back to back, ONE sync
        builds = []
        for b in range(B):
            for side in ("0", "1"):
                kp = data['keypoints' + side][b].to(torch.float32).contiguous()
                de = data['descriptors' + side][b].to(torch.float32).t().contiguous()      # (N, D) point-major
                n = kp.shape[0]
                if n < 2:
                    raise ValueError("need at least one array to concatenate")           # what the reference raises (agc.py:701)
                cap = n * 64
                work = torch.empty(hip.agc_workspace_bytes(n, D), dtype=torch.uint8, device=dev)
                kept = torch.empty(n, dtype=torch.int32, device=dev)
                indptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
                indices = torch.empty(cap, dtype=torch.int32, device=dev)
                info = torch.empty(8, dtype=torch.int32, device=dev)
                hip.agc_build(kp, de, radius, percentile, min_size, work, kept, indptr, indices, info)
                builds.append(dict(b=b, side=side, kp=kp, de=de, kept=kept, indptr=indptr, indices=indices, info=info, work=work))
        infos = torch.stack([g["info"] for g in builds]).cpu().numpy()                     # the one host sync of the build
        for g, inf in zip(builds, infos):
            if inf[7]:
                raise hip.GimsHipError("adaptive graph exceeded the edge capacity (64 directed edges per node)")
            g["n_kept"], g["n_edges"] = int(inf[0]), int(inf[1])
            if g["n_kept"] == 0:
                raise ValueError("need at least one array to concatenate")               # np.vstack([]) in agc.py:701
            g["kept"] = g["kept"][:g["n_kept"]]
            g["indptr"] = g["indptr"][:g["n_kept"] + 1]
            g["indices"] = g["indices"][:g["n_edges"]]
            g["info_host"] = inf

        # ---- kept-keypoint compaction + the reference's in-place dict mutation (gmatcher.py:244-252)
        row_off, off = [], 0
        for g in builds:
            row_off.append(off)
            off += g["n_kept"]
        n_tot = off
        feat = torch.empty((n_tot, D), dtype=torch.float32, device=dev)
        kpts_all = torch.empty((n_tot, 2), dtype=torch.float32, device=dev)
        graphs = {"0": [], "1": []}
        for g, ro in zip(builds, row_off):
            nk = g["n_kept"]
            hip.gather_rows(g["de"], g["kept"], feat[ro:ro + nk])
            idx = g["kept"].long()
            kpts_all[ro:ro + nk] = g["kp"][idx]
            sc = data['scores' + g["side"]][g["b"]][idx]
            g["rows"] = (ro, nk)
            graphs[g["side"]].append(GraphHandle(g["indptr"], g["indices"],
                                                 {"point": kpts_all[ro:ro + nk], "feat": feat[ro:ro + nk], "score": sc}))
        for side in ("0", "1"):
            gs = graphs[side]
            data['keypoints' + side] = torch.stack([h.ndata['point'] for h in gs])
            data['descriptors' + side] = torch.stack([h.ndata['feat'] for h in gs]).permute(0, 2, 1)
            data['scores' + side] = torch.stack([h.ndata['score'] for h in gs])
            data['kept_kpts%s_indices' % side] = [g["kept"].tolist() for g in builds if g["side"] == side]
            data['graph' + side] = gs

        # ---- GraphSAGE over the merged CSR of all images (gmatcher.py:145-162, 268-269)
        e_off = np.cumsum([0] + [g["n_edges"] for g in builds]).tolist()
        indptr_all = torch.cat([g["indptr"][:-1] + e0 for g, e0 in zip(builds, e_off[:-1])]
                               + [torch.tensor([e_off[-1]], dtype=torch.int32, device=dev)]).to(torch.int32)
        indices_all = torch.cat([g["indices"] + ro for g, ro in zip(builds, row_off)]).to(torch.int32)
        h = feat
        for i, e in enumerate(P["sage"]):
            agg = torch.empty_like(h)
            hip.sage_mean(h, indptr_all, indices_all, agg)
            h = self._lin(e, h, a1=agg, act=hip.ACT_RELU if i < 2 else hip.ACT_NONE)
        sage = h
        # ---- keypoint encoder (gmatcher.py:26-33, 87-97) ; desc = sage + kenc (gmatcher.py:270-271)
        norm3 = torch.empty((len(builds), 3), dtype=torch.float32)
        for i, g in enumerate(builds):
            shp = data['image' + g["side"]].shape
            height, width = shp[2], shp[3]                      # NHWC callers => (W, 3): the reference's quirk, kept verbatim
            one = torch.tensor(1, dtype=torch.float32)
            size = torch.stack([one * width, one * height])
            norm3[i, 0], norm3[i, 1] = size[0] / 2, size[1] / 2
            norm3[i, 2] = size.max() * 0.7
        seg = torch.cat([torch.full((g["n_kept"],), i, dtype=torch.int32) for i, g in enumerate(builds)]).to(dev)
        x = torch.empty((n_tot, P["kenc_w1"].shape[0]), dtype=torch.float32, device=dev)
        hip.kenc_first(kpts_all, norm3.to(dev), seg, P["kenc_w1"], P["kenc_b1"], x)
        for i, e in enumerate(P["kenc"]):
            last = i == len(P["kenc"]) - 1
            x = self._lin(e, x, act=hip.ACT_NONE if last else hip.ACT_RELU, residual=sage if last else None,
                          out=torch.empty((n_tot, e["n"]), dtype=torch.float32, device=dev))
        desc = x

        # ---- attentional GNN (gmatcher.py:99-143): per layer QKV -> flash attention -> merge -> MLP -> residual
        pairs = []
        for b in range(B):
            g0, g1 = builds[2 * b], builds[2 * b + 1]
            pairs.append((g0["rows"], g1["rows"]))
        self_pr = torch.tensor([[o, n, o, n] for pr in pairs for (o, n) in pr], dtype=torch.int32, device=dev)
        cross_pr = torch.tensor([q for (o0, n0), (o1, n1) in pairs for q in ((o0, n0, o1, n1), (o1, n1, o0, n0))],
                                dtype=torch.int32, device=dev)
        max_nq = max(g["n_kept"] for g in builds)
        qkv = torch.empty((n_tot, 3 * D), dtype=torch.bfloat16, device=dev)
        msg = torch.empty((n_tot, D), dtype=torch.float32, device=dev)
        mrg = torch.empty((n_tot, D), dtype=torch.float32, device=dev)
        hid = torch.empty((n_tot, 2 * D), dtype=torch.float32, device=dev)
        for L in P["layers"]:
            self._lin(L["qkv"], desc, out_bf16=qkv)
            hip.attention(qkv, cross_pr if L["cross"] else self_pr, max_nq, self._heads, msg, 0, D, 2 * D)
            self._lin(L["merge"], msg, out=mrg)
            self._lin(L["mlp0"], desc, a1=mrg, act=hip.ACT_RELU, out=hid)
            self._lin(L["mlp1"], hid, residual=desc, out=desc)          # desc += delta  (gmatcher.py:142)
        # ---- final projection, score matrix, Sinkhorn, selection (gmatcher.py:273-294)
        mdesc = self._lin(P["final"], desc)
        items, keep = [], []
        for (o0, n0), (o1, n1) in pairs:
            ld = (n1 + 3) // 4 * 4
            scores = torch.empty((n0, ld), dtype=torch.float32, device=dev)
            hip.linear(mdesc[o0:o0 + n0], mdesc[o1:o1 + n1], out=scores, precision=hip.PREC_F32, scale=1.0 / math.sqrt(D), n=n1)
            it = dict(scores=scores, n=n0, m=n1,
                      matches0=torch.empty(n0, dtype=torch.int64, device=dev), matches1=torch.empty(n1, dtype=torch.int64, device=dev),
                      mscores0=torch.empty(n0, dtype=torch.float32, device=dev), mscores1=torch.empty(n1, dtype=torch.float32, device=dev),
                      uv=torch.empty(n0 + n1 + 3, dtype=torch.float32, device=dev))
            items.append(it)
        probs = hip.make_ot_problems(items)
        work = torch.empty(hip.sinkhorn_workspace_bytes(probs), dtype=torch.uint8, device=dev)
        hip.sinkhorn_match(probs, P["alpha"], cfg['sinkhorn_iterations'], cfg['match_threshold'], work)
        self._last = dict(items=items, pairs=pairs, mdesc=mdesc, desc=desc, sage=sage, builds=builds)   # introspection for tests
        md0 = torch.stack([mdesc[o0:o0 + n0] for (o0, n0), _ in pairs])
        md1 = torch.stack([mdesc[o1:o1 + n1] for _, (o1, n1) in pairs])
        return {
            'keypoints0': data['keypoints0'], 'keypoints1': data['keypoints1'],
            'descriptors0': data['descriptors0'], 'descriptors1': data['descriptors1'],
            'matches0': torch.stack([it["matches0"] for it in items]),
            'matches1': torch.stack([it["matches1"] for it in items]),
            'matching_scores0': torch.stack([it["mscores0"] for it in items]),
            'matching_scores1': torch.stack([it["mscores1"] for it in items]),
            'mdesc0': md0.squeeze(), 'mdesc1': md1.squeeze(),
        }

"""CPU ORACLE for the GIMS matcher hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file restates, in plain NumPy + CPU PyTorch (no dgl / scipy / networkx / cv2), the
algorithm of the reference's matcher path:

    /root/reference/models/gmatcher.py   (GMatcher.forward and everything it calls)
    /root/reference/models/agc.py        (build_optimize_graph_with_cosine_similarity, live subset)

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the timed CPU baseline.  The product path (``gims_amd``) never imports
it and fails loudly when the HIP library is missing.

Pinning (how this oracle is known to equal the reference):
  * ``tools/gen_golden.py`` imports the real reference from /root/reference in the build container
    and commits golden input/output vectors to ``tests/golden/``; ``tests/test_oracle_golden.py``
    checks this file against them (edge sets / kept indices / match indices exact, floats <= 1e-5).
  * PARITY UNPINNED for the two third-party pieces whose source is absent from /root/reference:
    ``dgl.nn.SAGEConv(...,'mean')`` and ``dgl.from_networkx`` (dgl==1.1.2, requirements:16) --
    restated from DGL's documented semantics (see ``sage_conv_mean`` below); ``torch_scatter`` is
    train-only and not on this path.
  * Documented deviation: exact-distance ties in the sequential fix-ups (agc.py:476-495, 518-565) are
    broken by lowest node index here; SciPy's kd-tree order / CPython set order decide them in the
    reference.  Component centroids are accumulated in float64 here (float32 row-order in the
    reference, agc.py:542-543) -- only near-tied nearest-centroid decisions can differ.

Every function cites the reference lines it follows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

DEFAULT_CONFIG = {  # gmatcher.py:166-176
    "descriptor_dim": 256,
    "weights_path": None,
    "keypoint_encoder": [32, 64, 128, 256],
    "transformer_layers": ["self", "cross"] * 9,
    "sinkhorn_iterations": 100,
    "match_threshold": 0.2,
    "use_layernorm": False,
    "input_dim": 256,
    "num_heads": 4,
}
BN_EPS = 1e-5  # torch.nn.BatchNorm1d default (gmatcher.py:22)


# ============================================================================ AGC (models/agc.py)

def cosine_similarity_matrix(descs: np.ndarray) -> np.ndarray:
    """agc.py:382-391 -- F.normalize(dim=1) (eps 1e-12 clamp) then D D^T, float32 on CPU.

    ``descs`` is (N,D).  The reference hands in the *transposed view* of the (D,N) descriptor tensor
    (agc.py:431), and the BLAS path -- hence the last-ulp rounding of every similarity -- depends on
    that memory layout, so callers that want bit-equality with the reference pass the same view."""
    d = torch.from_numpy(descs).float()
    d = F.normalize(d, dim=1)
    return torch.matmul(d, d.T).numpy()


def percentile_index(length: int, percentile: float) -> int:
    """agc.py:378-379 -- k = int(L*p/100), clamped to L-1."""
    k = int(length * percentile / 100)
    if k >= length:
        k = length - 1
    return k


def percentile_threshold(sim: np.ndarray, percentile: float) -> np.float32:
    """agc.py:439-440 + 367-380 -- exact k-th smallest of the strict upper triangle."""
    vals = sim[np.triu_indices_from(sim, k=1)]
    k = percentile_index(len(vals), percentile)
    return np.partition(vals, k)[k]


def radius_pairs(kpts: np.ndarray, radius: float) -> np.ndarray:
    """agc.py:435-436 -- cKDTree.query_pairs(r): pairs i<j with ||xi-xj||^2 <= r^2, evaluated in
    float64 on the float32 coordinates, inclusive.  Returned sorted lexicographically, (E,2) int64."""
    p = kpts.astype(np.float64)
    n = len(p)
    r2 = float(radius) * float(radius)
    out = []
    step = 1024
    for a in range(0, n, step):
        pa = p[a:a + step]
        dx = pa[:, None, 0] - p[None, :, 0]
        dy = pa[:, None, 1] - p[None, :, 1]
        d2 = dx * dx + dy * dy
        ii, jj = np.nonzero(d2 <= r2)
        ii = ii + a
        m = ii < jj
        out.append(np.stack([ii[m], jj[m]], axis=1))
    e = np.concatenate(out, axis=0) if out else np.zeros((0, 2), np.int64)
    return e.astype(np.int64)


def coarse_graph(kpts: np.ndarray, descs: np.ndarray, radius: float, percentile: float):
    """agc.py:413-449 -- radius candidates filtered by sim >= percentile threshold."""
    sim = cosine_similarity_matrix(descs)
    thr = percentile_threshold(sim, percentile)
    cand = radius_pairs(kpts, radius)
    keep = sim[cand[:, 0], cand[:, 1]] >= thr
    return cand[keep], thr, sim, cand


class _Adj:
    """Tiny undirected simple-graph helper (adjacency sets), standing in for networkx.Graph."""

    def __init__(self, n: int, edges: np.ndarray):
        self.n = n
        self.adj: List[set] = [set() for _ in range(n)]
        self.m = 0
        for u, v in edges:
            self.add(int(u), int(v))

    def add(self, u: int, v: int):
        if u == v:
            # networkx would add a self-loop; the callers never produce one
            return
        if v not in self.adj[u]:
            self.adj[u].add(v)
            self.adj[v].add(u)
            self.m += 1


def _nearest_other(p64: np.ndarray, i: int) -> int:
    d = p64 - p64[i]
    d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]
    d2[i] = np.inf
    return int(np.argmin(d2))  # ties -> lowest index (documented deviation)


def connect_isolated_nodes(g: _Adj, kpts: np.ndarray):
    """agc.py:476-495 -- in ascending node order, every node whose *current* degree is 0 gets an edge
    to its nearest other node.  No-op when the graph has no nodes or no edges (agc.py:486)."""
    if g.n == 0 or g.m == 0:
        return g
    p64 = kpts.astype(np.float64)
    for node in range(g.n):
        if len(g.adj[node]) == 0:
            g.add(node, _nearest_other(p64, node))
    return g


def connected_components(n: int, adj: Sequence[set], alive: np.ndarray) -> List[List[int]]:
    """networkx.connected_components order: components by their lowest alive node id."""
    seen = np.zeros(n, dtype=bool)
    comps = []
    for s in range(n):
        if not alive[s] or seen[s]:
            continue
        stack = [s]
        seen[s] = True
        comp = []
        while stack:
            u = stack.pop()
            comp.append(u)
            for v in adj[u]:
                if not seen[v]:
                    seen[v] = True
                    stack.append(v)
        comps.append(sorted(comp))
    return comps


def remove_small_components(g: _Adj, min_size: int):
    """agc.py:497-516 -- drop every component with fewer than min_size nodes."""
    alive = np.ones(g.n, dtype=bool)
    for comp in connected_components(g.n, g.adj, alive):
        if len(comp) < min_size:
            alive[comp] = False
    return alive


def fast_connect_components(g: _Adj, alive: np.ndarray, kpts: np.ndarray):
    """agc.py:518-565 -- ONE round: each component links to its nearest-centroid component through
    the closest node pair; (i,j)/(j,i) de-duplicated in component order.  Returns the added edges."""
    comps = connected_components(g.n, g.adj, alive)
    added = []
    if len(comps) <= 1:
        return added
    p64 = kpts.astype(np.float64)
    cent = np.stack([p64[c].mean(axis=0) for c in comps])
    done = set()
    for i in range(len(comps)):
        d = cent - cent[i]
        d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]
        d2[i] = np.inf
        j = int(np.argmin(d2))                      # tree.query(k=2)[:,1]  (agc.py:547)
        if (i, j) in done or (j, i) in done:
            continue
        done.add((i, j))
        ci, cj = np.asarray(comps[i]), np.asarray(comps[j])
        dx = p64[cj][:, None, 0] - p64[ci][None, :, 0]
        dy = p64[cj][:, None, 1] - p64[ci][None, :, 1]
        dd = dx * dx + dy * dy                       # (|cj|, |ci|)
        best_i = np.argmin(dd, axis=1)               # nearest node of comp i for every node of comp j
        best_d = dd[np.arange(len(cj)), best_i]
        jj = int(np.argmin(best_d))                  # agc.py:560
        u, v = int(ci[best_i[jj]]), int(cj[jj])
        g.add(u, v)
        added.append((u, v))
    return added


def agc_build(kpts: np.ndarray, descs: np.ndarray, radius=20, percentile=50, min_size=10) -> Dict:
    """agc.py:682-709 for ONE image.  kpts (N,2) f32, descs (N,D) f32 (point-major).

    Returns kept (sorted original ids), the final undirected edge list in *kept-relabelled* ids
    (u<v, lexicographically sorted), the CSR of the bidirectional DGL graph (dgl.from_networkx,
    agc.py:704), the threshold and the stage edge lists (original ids) for stage-wise tests."""
    n = len(kpts)
    coarse, thr, sim, cand = coarse_graph(kpts, descs, radius, percentile)
    g = _Adj(n, coarse)
    m_coarse = g.m
    connect_isolated_nodes(g, kpts)
    after_iso = _edge_list(g, np.ones(n, dtype=bool))
    alive = remove_small_components(g, min_size)
    kept = np.nonzero(alive)[0].astype(np.int64)
    if len(kept) == 0:
        # agc.py:701 np.vstack([]) raises ValueError in the reference
        raise ValueError("need at least one array to concatenate")
    added = fast_connect_components(g, alive, kpts)
    final_orig = _edge_list(g, alive)
    relabel = -np.ones(n, dtype=np.int64)
    relabel[kept] = np.arange(len(kept))
    e = relabel[final_orig]
    indptr, indices = csr_from_undirected(len(kept), e)
    return {"kept": kept, "edges": e, "indptr": indptr, "indices": indices, "threshold": np.float32(thr),
            "coarse_edges": _sorted_edges(coarse), "iso_edges": after_iso, "final_edges_orig": final_orig,
            "link_edges": np.asarray(added, dtype=np.int64).reshape(-1, 2), "n_coarse": m_coarse,
            "n_candidates": len(cand)}


def _sorted_edges(e: np.ndarray) -> np.ndarray:
    e = np.asarray(e, dtype=np.int64).reshape(-1, 2)
    e = np.stack([e.min(axis=1), e.max(axis=1)], axis=1)
    order = np.lexsort((e[:, 1], e[:, 0]))
    return e[order]


def _edge_list(g: _Adj, alive: np.ndarray) -> np.ndarray:
    out = [(u, v) for u in range(g.n) if alive[u] for v in g.adj[u] if u < v and alive[v]]
    return _sorted_edges(np.asarray(out, dtype=np.int64).reshape(-1, 2))


def csr_from_undirected(n: int, edges: np.ndarray):
    """dgl.from_networkx (agc.py:704): both directions; CSR keyed by destination, neighbours ascending."""
    src = np.concatenate([edges[:, 0], edges[:, 1]])
    dst = np.concatenate([edges[:, 1], edges[:, 0]])
    order = np.lexsort((src, dst))
    src, dst = src[order], dst[order]
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(indptr, dst + 1, 1)
    indptr = np.cumsum(indptr)
    return indptr.astype(np.int32), src.astype(np.int32)


# ============================================================================ GMatcher modules

def _t(sd, name) -> torch.Tensor:
    v = sd[name]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def sage_bias(sd, i: int) -> torch.Tensor:
    """SAGEConv bias: DGL >= 1.0 keeps it on fc_self, older DGL as a separate parameter."""
    p = f"gnn_encoder.layers.{i}."
    if p + "fc_self.bias" in sd:
        return _t(sd, p + "fc_self.bias")
    return _t(sd, p + "bias")


def sage_conv_mean(sd, i: int, indptr: np.ndarray, indices: np.ndarray, h: torch.Tensor) -> torch.Tensor:
    """dgl.nn.SAGEConv(in,out,'mean') (call sites gmatcher.py:149-151,158) -- restated from DGL docs:
    rst = fc_self(h) + fc_neigh(mean of in-neighbour h); fc_neigh before the mean iff in>out."""
    p = f"gnn_encoder.layers.{i}."
    w_self, w_neigh, b = _t(sd, p + "fc_self.weight"), _t(sd, p + "fc_neigh.weight"), sage_bias(sd, i)
    out_f, in_f = w_self.shape
    lin_before = in_f > out_f
    src = torch.from_numpy(indices.astype(np.int64))
    deg = torch.from_numpy(np.diff(indptr).astype(np.int64))
    dst = torch.repeat_interleave(torch.arange(len(deg)), deg)
    x = F.linear(h, w_neigh) if lin_before else h
    agg = torch.zeros(len(deg), x.shape[1], dtype=h.dtype)
    agg.index_add_(0, dst, x[src])
    agg = agg / deg.clamp(min=1).to(h.dtype).unsqueeze(1)
    if not lin_before:
        agg = F.linear(agg, w_neigh)
    return F.linear(h, w_self, b) + agg


def graph_sage(sd, indptr, indices, feat: torch.Tensor) -> torch.Tensor:
    """gmatcher.py:145-162 -- 3 SAGEConv layers, ReLU after all but the last."""
    h = feat
    for i in range(3):
        h = sage_conv_mean(sd, i, indptr, indices, h)
        if i != 2:
            h = F.relu(h)
    return h


def normalize_keypoints(kpts: torch.Tensor, image_shape) -> torch.Tensor:
    """gmatcher.py:26-33 -- NOTE the reference unpacks the caller's NHWC shape as NCHW."""
    _, _, height, width = image_shape
    one = kpts.new_tensor(1)
    size = torch.stack([one * width, one * height])[None]
    center = size / 2
    scaling = size.max(1, keepdim=True).values * 0.7
    return (kpts - center[:, None, :]) / scaling[:, None, :]


def _mlp(sd, prefix: str, n_convs: int, x: torch.Tensor, bn_training: bool = False) -> torch.Tensor:
    """gmatcher.py:11-24 -- Conv1d(k=1) [+ BatchNorm1d | LayerNorm + ReLU] stack; Sequential indices 0,1,2 / 3,4,5 ...
    Which norm a checkpoint uses is read off its keys (LayerNorm stores a_2 / b_2, gmatcher.py:78-79).  ``bn_training``:
    nn.BatchNorm1d in train() mode -- statistics of THIS call over (B, N), running statistics (torch tensors in ``sd``) updated
    in place with momentum 0.1 and the unbiased variance, num_batches_tracked incremented when present."""
    idx = 0
    for i in range(n_convs):
        x = F.conv1d(x, _t(sd, f"{prefix}.{idx}.weight"), _t(sd, f"{prefix}.{idx}.bias"))
        idx += 1
        if i < n_convs - 1:
            if f"{prefix}.{idx}.a_2" in sd:
                # use_layernorm=True (gmatcher.py:19-20, 74-85): over the CHANNEL dim of (B,C,N), unbiased std, eps added to std
                mean = x.mean(-2, keepdim=True)
                std = x.std(-2, keepdim=True)
                x = _t(sd, f"{prefix}.{idx}.a_2").reshape(1, -1, 1) * ((x - mean) / (std + 1e-6)) + _t(sd, f"{prefix}.{idx}.b_2").reshape(1, -1, 1)
            else:
                x = F.batch_norm(x, _t(sd, f"{prefix}.{idx}.running_mean"), _t(sd, f"{prefix}.{idx}.running_var"),
                                 _t(sd, f"{prefix}.{idx}.weight"), _t(sd, f"{prefix}.{idx}.bias"),
                                 training=bn_training, momentum=0.1, eps=BN_EPS)
                if bn_training and isinstance(sd.get(f"{prefix}.{idx}.num_batches_tracked"), torch.Tensor):
                    sd[f"{prefix}.{idx}.num_batches_tracked"] += 1
            x = F.relu(x)
            idx += 2
    return x


def keypoint_encoder(sd, kpts_norm: torch.Tensor, n_convs: int = 5, bn_training: bool = False) -> torch.Tensor:
    """gmatcher.py:87-97 with score=False: encoder(kpts^T) -> (B,256,N)."""
    return _mlp(sd, "kenc.encoder", n_convs, kpts_norm.transpose(1, 2), bn_training)


def attention(q, k, v):
    """gmatcher.py:35-39."""
    dim = q.shape[1]
    scores = torch.einsum("bdhn,bdhm->bhnm", q, k) / dim ** 0.5
    prob = F.softmax(scores, dim=-1)
    return torch.einsum("bhnm,bdhm->bdhn", prob, v)


def attentional_propagation(sd, l: int, x: torch.Tensor, source: torch.Tensor, heads: int = 4, bn_training: bool = False) -> torch.Tensor:
    """gmatcher.py:99-125 -- MultiHeadedAttention (heads interleaved: view(B, dh, H, N)) + MLP([2D,2D,D])."""
    p = f"gnn.layers.{l}."
    b, d, _ = x.shape
    q, k, v = [F.conv1d(t, _t(sd, p + f"attn.proj.{j}.weight"), _t(sd, p + f"attn.proj.{j}.bias"))
               .view(b, d // heads, heads, -1) for j, t in enumerate((x, source, source))]
    msg = attention(q, k, v).contiguous().view(b, d, -1)
    msg = F.conv1d(msg, _t(sd, p + "attn.merge.weight"), _t(sd, p + "attn.merge.bias"))
    return _mlp(sd, p + "mlp", 2, torch.cat([x, msg], dim=1), bn_training)


def attentional_gnn(sd, desc0, desc1, names, taps=None, bn_training: bool = False):
    """gmatcher.py:127-143 (image 0 before image 1 in every layer: the order of the two running-statistics updates)."""
    for l, name in enumerate(names):
        if name == "cross":
            src0, src1 = desc1, desc0
        else:
            src0, src1 = desc0, desc1
        delta0 = attentional_propagation(sd, l, desc0, src0, bn_training=bn_training)
        delta1 = attentional_propagation(sd, l, desc1, src1, bn_training=bn_training)
        desc0, desc1 = desc0 + delta0, desc1 + delta1
        if taps is not None:
            taps.append((desc0.clone(), desc1.clone()))
    return desc0, desc1


def log_sinkhorn_iterations(Z, log_mu, log_nu, iters: int):
    """gmatcher.py:41-47."""
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(Z + u.unsqueeze(2), dim=1)
    return Z + u.unsqueeze(2) + v.unsqueeze(1)


def log_optimal_transport(scores, alpha, iters: int):
    """gmatcher.py:49-69."""
    b, m, n = scores.shape
    one = scores.new_tensor(1)
    ms, ns = (m * one).to(scores), (n * one).to(scores)
    bins0 = alpha.expand(b, m, 1)
    bins1 = alpha.expand(b, 1, n)
    alpha = alpha.expand(b, 1, 1)
    couplings = torch.cat([torch.cat([scores, bins0], -1), torch.cat([bins1, alpha], -1)], 1)
    norm = -(ms + ns).log()
    log_mu = torch.cat([norm.expand(m), ns.log()[None] + norm])
    log_nu = torch.cat([norm.expand(n), ms.log()[None] + norm])
    log_mu, log_nu = log_mu[None].expand(b, -1), log_nu[None].expand(b, -1)
    Z = log_sinkhorn_iterations(couplings, log_mu, log_nu, iters)
    return Z - norm


def select_matches(scores: torch.Tensor, match_threshold: float):
    """gmatcher.py:284-294 -- mutual argmax on the inner block, exp(max) > threshold, -1 fill."""
    max0, max1 = scores[:, :-1, :-1].max(2), scores[:, :-1, :-1].max(1)
    indices0, indices1 = max0.indices, max1.indices
    ar0 = torch.arange(indices0.shape[1])[None]
    ar1 = torch.arange(indices1.shape[1])[None]
    mutual0 = ar0 == indices1.gather(1, indices0)
    mutual1 = ar1 == indices0.gather(1, indices1)
    zero = scores.new_tensor(0)
    mscores0 = torch.where(mutual0, max0.values.exp(), zero)
    mscores1 = torch.where(mutual1, mscores0.gather(1, indices1), zero)
    valid0 = mutual0 & (mscores0 > match_threshold)
    valid1 = mutual1 & valid0.gather(1, indices1)
    indices0 = torch.where(valid0, indices0, indices0.new_tensor(-1))
    indices1 = torch.where(valid1, indices1, indices1.new_tensor(-1))
    return indices0, indices1, mscores0, mscores1


# ============================================================================ GMatcher.forward

def train_loss(ot: torch.Tensor, matches: torch.Tensor, kept0, kept1, batch_size: int, pos_w: float, neg_w: float):
    """gmatcher.py:333-386 (forward_train after the Sinkhorn solve): ``ot`` (B, N+1, M+1) log-OT matrix, ``matches`` (K, 3)
    int64 rows (b, i0, i1) in ORIGINAL keypoint ids, kept0/kept1 per batch element the sorted kept ids.  Restates the
    remap (340-367), the negative-index gather that reads the corner cell for (b, -1, -1) rows (372), the clamp (374), and
    ``torch_scatter.scatter_mean`` (380; third-party, absent from /root/reference and not installed: restated from its
    documented semantics -- mean per index, 0 for empty groups -- PARITY UNPINNED for that call)."""
    remap0 = [{int(o): i for i, o in enumerate(k)} for k in kept0]
    remap1 = [{int(o): i for i, o in enumerate(k)} for k in kept1]
    rows = []
    for b, i0, i1 in matches.tolist():
        if i0 == -1 or i1 == -1 or i0 not in remap0[b] or i1 not in remap1[b]:
            rows.append([b, -1, -1])
        else:
            rows.append([b, remap0[b][i0], remap1[b][i1]])
    gt = torch.tensor(rows, dtype=torch.long).reshape(-1, 3)
    neg = (gt[:, 1] == -1) | (gt[:, 2] == -1)
    vec = -torch.clamp(ot[gt[:, 0], gt[:, 1], gt[:, 2]], min=-100, max=0.0)

    def scatter_mean(src, index):
        out = torch.zeros(batch_size, dtype=src.dtype).index_add_(0, index, src)
        cnt = torch.zeros(batch_size, dtype=src.dtype).index_add_(0, index, torch.ones_like(src))
        return out / cnt.clamp(min=1)

    pos_loss = pos_w * scatter_mean(vec[~neg], gt[:, 0][~neg]).mean()
    neg_loss = neg_w * scatter_mean(vec[neg], gt[:, 0][neg]).mean()
    return pos_loss + neg_loss, pos_loss, neg_loss


def gmatcher_forward(sd, data: dict, config: dict | None = None, stages: dict | None = None, mode: str = "test"):
    """gmatcher.py:219-307 (test mode) on a state dict ``sd`` (NumPy arrays or tensors); ``mode='train'`` returns the
    forward value of forward_train's loss instead (gmatcher.py:254, 309-386; eval-mode BatchNorm unless
    ``config['bn_training']``).

    ``data`` holds torch tensors in the reference layout and is mutated in place exactly
    like the reference does (gmatcher.py:244-252).  ``stages`` (optional dict) receives intermediates.
    """
    cfg = {**DEFAULT_CONFIG, **(config or {})}
    radius = data.get("radius", 25)
    percentile = data.get("percentile", 7)
    min_size = data.get("min_size", 8)
    graphs = []
    for side in ("0", "1"):
        kp, de, sc = data["keypoints" + side], data["descriptors" + side], data["scores" + side]
        per_b = []
        for b in range(kp.shape[0]):
            kp_np = kp[b].detach().cpu().numpy()
            de_np = de[b].permute(1, 0).detach().cpu().numpy()        # transposed view, as agc.py:431
            g = agc_build(kp_np, de_np, radius, percentile, min_size)
            idx = torch.from_numpy(g["kept"])
            g["point"], g["feat"], g["score"] = kp[b][idx], de[b].permute(1, 0)[idx].contiguous(), sc[b][idx]
            per_b.append(g)
        graphs.append(per_b)
    g0s, g1s = graphs
    data["keypoints0"] = torch.stack([g["point"] for g in g0s])
    data["descriptors0"] = torch.stack([g["feat"] for g in g0s]).permute(0, 2, 1)
    data["keypoints1"] = torch.stack([g["point"] for g in g1s])
    data["descriptors1"] = torch.stack([g["feat"] for g in g1s]).permute(0, 2, 1)
    data["scores0"] = torch.stack([g["score"] for g in g0s])
    data["scores1"] = torch.stack([g["score"] for g in g1s])
    data["kept_kpts0_indices"] = [g["kept"].tolist() for g in g0s]
    data["kept_kpts1_indices"] = [g["kept"].tolist() for g in g1s]
    data["graph0"], data["graph1"] = g0s, g1s

    kpts0, kpts1 = data["keypoints0"], data["keypoints1"]
    if kpts0.shape[1] == 0 or kpts1.shape[1] == 0:  # gmatcher.py:257-264
        shape0, shape1 = kpts0.shape[:-1], kpts1.shape[:-1]
        return {"matches0": kpts0.new_full(shape0, -1, dtype=torch.int),
                "matches1": kpts1.new_full(shape1, -1, dtype=torch.int),
                "matching_scores0": kpts0.new_zeros(shape0), "matching_scores1": kpts1.new_zeros(shape1)}
    kn0 = normalize_keypoints(kpts0, data["image0"].shape)
    kn1 = normalize_keypoints(kpts1, data["image1"].shape)
    sage0 = torch.stack([graph_sage(sd, g["indptr"], g["indices"], g["feat"]) for g in g0s]).permute(0, 2, 1)
    sage1 = torch.stack([graph_sage(sd, g["indptr"], g["indices"], g["feat"]) for g in g1s]).permute(0, 2, 1)
    n_kenc = len(cfg["keypoint_encoder"]) + 1
    bn_tr = bool(cfg.get("bn_training", False))          # nn.Module.train(): BatchNorm on batch statistics (train.py:100)
    ke0, ke1 = keypoint_encoder(sd, kn0, n_kenc, bn_tr), keypoint_encoder(sd, kn1, n_kenc, bn_tr)
    desc0, desc1 = sage0 + ke0, sage1 + ke1
    taps = [] if stages is not None else None
    gd0, gd1 = attentional_gnn(sd, desc0, desc1, cfg["transformer_layers"], taps, bn_tr)
    fw, fb = _t(sd, "final_proj.weight"), _t(sd, "final_proj.bias")
    mdesc0, mdesc1 = F.conv1d(gd0, fw, fb), F.conv1d(gd1, fw, fb)
    scores = torch.einsum("bdn,bdm->bnm", mdesc0, mdesc1)
    scores = scores / cfg["descriptor_dim"] ** 0.5
    alpha = _t(sd, "bin_score").float()
    if stages is not None and stages.get("grad_scores"):        # leaves for autograd: d loss / d scores, d loss / d bin_score
        scores = scores.detach().requires_grad_(True)
        alpha = alpha.detach().clone().requires_grad_(True)
        stages["scores_leaf"], stages["alpha_leaf"] = scores, alpha
    ot = log_optimal_transport(scores, alpha, iters=cfg["sinkhorn_iterations"])
    if mode == "train":
        return train_loss(ot, data["matches"], data["kept_kpts0_indices"], data["kept_kpts1_indices"], data["image0"].shape[0],
                          cfg["pos_loss_weight"], cfg["neg_loss_weight"])
    i0, i1, s0, s1 = select_matches(ot, cfg["match_threshold"])
    if stages is not None:
        stages.update(kn0=kn0, kn1=kn1, sage0=sage0, sage1=sage1, kenc0=ke0, kenc1=ke1, desc0_in=desc0,
                      desc1_in=desc1, gnn_taps=taps, gnn0=gd0, gnn1=gd1, scores=scores, ot=ot)
    return {"keypoints0": data["keypoints0"], "keypoints1": data["keypoints1"],
            "descriptors0": data["descriptors0"], "descriptors1": data["descriptors1"],
            "matches0": i0, "matches1": i1, "matching_scores0": s0, "matching_scores1": s1,
            "mdesc0": mdesc0.permute(0, 2, 1).squeeze(), "mdesc1": mdesc1.permute(0, 2, 1).squeeze()}


def train_step(sd_np: dict, data: dict, config: dict | None = None):
    """One step of train.py:100, 136-137 on the CPU: ``model.train()``, forward(mode='train'), ``loss.backward()`` -- torch
    autograd through this file's restatement.  Returns (loss, pos, neg) as floats, the gradient of every floating-point
    parameter (NumPy, keyed like the state dict) and the state dict's BatchNorm buffers after the step."""
    sd = {}
    for k, v in sd_np.items():
        t = torch.from_numpy(np.array(v, copy=True))
        is_buf = k.endswith(("running_mean", "running_var", "num_batches_tracked"))
        sd[k] = t if is_buf else t.requires_grad_(True)
    with torch.enable_grad():
        loss, pos, neg = gmatcher_forward(sd, data, {**(config or {}), "bn_training": True}, mode="train")
        loss.backward()
    grads = {k: v.grad.numpy() for k, v in sd.items() if v.requires_grad and v.grad is not None}
    bufs = {k: v.detach().numpy() for k, v in sd.items() if not v.requires_grad}
    return (float(loss.detach()), float(pos.detach()), float(neg.detach())), grads, bufs

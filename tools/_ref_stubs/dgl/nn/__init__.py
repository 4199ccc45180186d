"""dgl.nn.SAGEConv('mean') restated from DGL 1.1 documented behaviour (stub, see ../__init__.py).

rst = fc_self(h_dst) + fc_neigh(mean_{j in in-neighbours(i)} h_j)
  * fc_neigh is applied BEFORE aggregation iff in_feats > out_feats (lin_before_mp);
  * nodes with in-degree 0 get a zero neighbour term;
  * DGL >= 1.0: bias lives on fc_self (fc_neigh has no bias).
"""
import torch
import torch.nn as tnn


class SAGEConv(tnn.Module):
    def __init__(self, in_feats, out_feats, aggregator_type, feat_drop=0.0, bias=True, norm=None, activation=None):
        super().__init__()
        assert aggregator_type == "mean"
        self._in, self._out = in_feats, out_feats
        self.fc_neigh = tnn.Linear(in_feats, out_feats, bias=False)
        self.fc_self = tnn.Linear(in_feats, out_feats, bias=bias)

    def forward(self, graph, feat):
        src, dst = graph.edges()
        n = graph.num_nodes()
        lin_before_mp = self._in > self._out
        h = self.fc_neigh(feat) if lin_before_mp else feat
        agg = torch.zeros(n, h.shape[1], dtype=h.dtype, device=h.device)
        agg.index_add_(0, dst, h[src])
        deg = torch.zeros(n, dtype=h.dtype, device=h.device)
        deg.index_add_(0, dst, torch.ones(dst.numel(), dtype=h.dtype, device=h.device))
        h_neigh = agg / deg.clamp(min=1).unsqueeze(1)
        if not lin_before_mp:
            h_neigh = self.fc_neigh(h_neigh)
        return self.fc_self(feat) + h_neigh

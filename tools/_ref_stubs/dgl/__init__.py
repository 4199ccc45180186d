"""Stand-in for dgl==1.1.2 (requirements:16), used ONLY by tools/gen_golden.py in the build
container -- DGL is not installed and there is no network.

It restates the two DGL entry points the matcher hot path touches, from DGL's documented
semantics (the DGL source is NOT available offline, so this part of the oracle is
"parity unpinned" against a real DGL wheel -- see DESIGN.md):

* dgl.from_networkx(nx_graph, device=...)  (agc.py:704): nodes relabelled to 0..n-1 in sorted
  order of their ids, every undirected edge stored in both directions.
* dgl.nn.SAGEConv(in, out, 'mean')          (gmatcher.py:149-151,158): see dgl/nn/__init__.py.
"""
import numpy as np
import torch
from . import nn  # noqa: F401


class DGLGraph:
    def __init__(self, num_nodes, src, dst, device):
        self._n = int(num_nodes)
        self.src = torch.as_tensor(src, dtype=torch.int64, device=device)
        self.dst = torch.as_tensor(dst, dtype=torch.int64, device=device)
        self.ndata = {}
        self.device = device

    def num_nodes(self):
        return self._n

    number_of_nodes = num_nodes

    def num_edges(self):
        return int(self.src.numel())

    def edges(self):
        return self.src, self.dst

    def to(self, device):
        g = DGLGraph(self._n, self.src, self.dst, device)
        g.ndata = {k: v.to(device) for k, v in self.ndata.items()}
        return g


def from_networkx(nx_graph, device=None, **_):
    nodes = sorted(nx_graph.nodes)
    remap = {n: i for i, n in enumerate(nodes)}
    src, dst = [], []
    for u, v in nx_graph.edges:
        a, b = remap[u], remap[v]
        src += [a, b]
        dst += [b, a]
    return DGLGraph(len(nodes), np.asarray(src, dtype=np.int64), np.asarray(dst, dtype=np.int64),
                    device if device is not None else torch.device("cpu"))

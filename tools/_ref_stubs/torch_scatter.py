"""Stand-in for torch_scatter, used ONLY by tools/gen_golden.py (build container).
models/gmatcher.py imports it at module scope (gmatcher.py:6); scatter_mean is train-only
(gmatcher.py:380)."""
import torch


def scatter_mean(src, index, dim=0, dim_size=None):
    n = int(dim_size) if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
    out = torch.zeros(n, dtype=src.dtype, device=src.device)
    cnt = torch.zeros(n, dtype=src.dtype, device=src.device)
    out.index_add_(0, index, src)
    cnt.index_add_(0, index, torch.ones_like(src))
    return out / cnt.clamp(min=1)

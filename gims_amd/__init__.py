"""gims_amd -- the GIMS matcher hot path (models/gmatcher.py + models/agc.py of songxf1024/GIMS) on MI355X.

Public surface mirrors the reference: ``GMatcher(config).forward(data)`` and ``Matching(config).forward(data)``.
Kernels live in ``gims_amd/csrc`` (HIP, gfx950) behind the C ABI of ``include/gims_hip.h``.
"""
from .gmatcher import GMatcher  # noqa: F401
from .matching import Matching  # noqa: F401

__all__ = ["GMatcher", "Matching"]

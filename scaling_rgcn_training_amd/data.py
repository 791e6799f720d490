"""Minimal stand-in for ``torch_geometric.data.Data`` as the reference uses it: an attribute bag
created as ``Data(edge_index=...)`` (/root/reference/graphs/graph.py:68-69), extended by plain
attribute assignment (graphs/dataset.py:30-35,53-54) and moved with ``.to(device)``
(model/modelTrainer.py:43)."""
from __future__ import annotations

import torch


class Data:
    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k in self.__dict__ if not k.startswith("_")]

    def to(self, device, non_blocking: bool = False) -> "Data":
        out = Data()
        for k, v in self.__dict__.items():
            setattr(out, k, v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v)
        return out

    def __repr__(self):
        parts = []
        for k in self.keys():
            v = getattr(self, k)
            parts.append(f"{k}={list(v.shape)}" if torch.is_tensor(v) else f"{k}={v!r}")
        return "Data(" + ", ".join(parts) + ")"

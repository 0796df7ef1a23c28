"""FeatureDict / NeighborList: engine-backed views with the reference's field names.

* ``NeighborList`` exposes what callers of jax-md's NeighborList read: ``idx`` ((2,E_cap) int32,
  row 0 receivers / row 1 senders, padding = N - case_setup/features.py:110) and
  ``did_buffer_overflow`` (evaluate/rollout.py:135,140), plus ``update``.
* ``FeatureDict`` has the keys of lagrangebench/case_setup/features.py:47-126.  Values are
  exported from the engine lazily: the fused GNS path never materialises them, a user model that
  reads ``features["vel_hist"]`` gets them on first access.
"""
from __future__ import annotations

from typing import Dict, Iterator, Optional

import torch

from .. import _lib

_NODE_KEYS = ("vel_hist", "vel_mag", "bound", "force")
_EDGE_KEYS = ("rel_disp", "rel_dist")
_IDX_KEYS = ("senders", "receivers")


class _StaleError(RuntimeError):
    pass


class NeighborList:
    def __init__(self, engine, batched: bool, _snapshot=None):
        self.engine = engine
        self.batched = batched
        self.version = engine.version
        self.cell_capacity = engine.cell_capacity
        self.max_occupancy = engine.e_cap
        self._idx: Optional[torch.Tensor] = None
        self._n_edges: Optional[torch.Tensor] = None
        flags = engine.nl_flags().to(torch.bool)
        self.did_buffer_overflow = flags if batched else flags[0]

    def _export(self):
        if self._idx is None:
            if self.engine.version != self.version:
                raise _StaleError("NeighborList.idx requested after the engine state moved on; read it "
                                  "right after allocate/preprocess")
            idx, ne = self.engine.nl_idx()
            self._idx, self._n_edges = (idx, ne) if self.batched else (idx[0], ne[0])

    @property
    def idx(self) -> torch.Tensor:
        self._export()
        return self._idx

    @property
    def n_edges(self) -> torch.Tensor:
        self._export()
        return self._n_edges

    def update(self, position=None, **kwargs) -> "NeighborList":
        """neighbors.update(position): rebuild with the frozen capacities for the engine's current
        window (``position`` is accepted for signature parity and must be that window's newest frame)."""
        eng = self.engine
        if (eng.e_cap, eng.cell_capacity) != (self.max_occupancy, self.cell_capacity):
            eng.nl_set_capacity(self.cell_capacity, self.max_occupancy)
        eng.nl_update()
        return NeighborList(eng, self.batched)

    # pytree hook for utils.broadcast_to_batch / broadcast_from_batch (utils.py:38-47): capacities
    # are what matters when a list is re-used for another batch.
    def _lb_tree_map(self, fn):
        return self


class FeatureDict(dict):
    def __init__(self, engine, abs_pos: torch.Tensor, batched: bool):
        super().__init__()
        self.engine = engine
        self.batched = batched
        self.version = engine.version
        dict.__setitem__(self, "abs_pos", abs_pos if batched else abs_pos[0])
        self._keys = ["abs_pos", "vel_hist"]
        if engine.has_vel_mag:
            self._keys.append("vel_mag")
        if engine.has_bound:
            self._keys.append("bound")
        if engine.desc.force_kind != _lib.LB_FORCE_NONE:
            self._keys.append("force")
        self._keys += ["senders", "receivers", "rel_disp", "rel_dist"]

    # ---- lazy materialisation --------------------------------------------------------
    def _check(self, key):
        if self.engine.version != self.version:
            raise _StaleError(f"features[{key!r}] requested after the engine state moved on")

    def _put(self, d: Dict[str, torch.Tensor]):
        for k, v in d.items():
            dict.__setitem__(self, k, v if self.batched else v[0])

    def __missing__(self, key):
        if key not in self._keys:
            raise KeyError(key)
        self._check(key)
        if key in _NODE_KEYS:
            self._put(self.engine.node_features())
        elif key in _EDGE_KEYS:
            self._put(self.engine.edge_features())
        elif key in _IDX_KEYS:
            idx, _ = self.engine.nl_idx()
            self._put({"receivers": idx[:, 0], "senders": idx[:, 1]})
        return dict.__getitem__(self, key)

    def __contains__(self, key) -> bool:
        return key in self._keys

    def keys(self):
        return list(self._keys)

    def __iter__(self) -> Iterator[str]:
        return iter(self._keys)

    def __len__(self) -> int:
        return len(self._keys)

    def items(self):
        return [(k, self[k]) for k in self._keys]

    def values(self):
        return [self[k] for k in self._keys]

    def get(self, key, default=None):
        return self[key] if key in self._keys else default

    def materialize(self) -> "FeatureDict":
        for k in self._keys:
            self[k]
        return self

"""ConditionerWrapper: runs a list of conditioners over one batch and merges what they return.

Contract (src/lbm/models/embedders/conditioners_wrapper.py:16-114):
  * state-dict prefix ``conditioners.<i>.``;
  * result ``{"cond": {kind: tensor}}``; two conditioners emitting the same kind are concatenated in list
    order — "vector" and "concat" along dim 1, "crossattn" along dim 2;
  * dropping: a conditioner whose ``input_key`` is in ``ucg_keys`` is always asked for the dropped embedding
    (no random draw); otherwise, when its ``ucg_rate`` is positive and ``set_ucg_rate_zero`` is false, ONE
    ``torch.rand(1)`` from the CPU generator decides (drop when the draw is below the rate) — the draw belongs to
    the reproducible random stream of a training step, so it is made exactly then and never otherwise.
"""
from typing import Any, Dict, Iterable, List, Optional

import torch
import torch.nn as nn

from ..base.base_model import PlacementTracking

KEY2CATDIM = {"vector": 1, "crossattn": 2, "concat": 1}


class ConditionerWrapper(PlacementTracking, nn.Module):
    def __init__(self, conditioners: Optional[List[nn.Module]] = None):
        nn.Module.__init__(self)
        self._init_placement()
        self.conditioners = nn.ModuleList(conditioners)

    @staticmethod
    def _wants_drop(conditioner, forced_keys: Iterable[str], rates_off: bool) -> bool:
        if conditioner.input_key in forced_keys:
            return True
        if conditioner.ucg_rate > 0 and not rates_off:
            return bool(torch.rand(1) < conditioner.ucg_rate)
        return False

    @staticmethod
    def _merge(merged: Dict[str, torch.Tensor], new: Dict[str, torch.Tensor]) -> None:
        for kind, value in new.items():
            merged[kind] = value if kind not in merged else torch.cat([merged[kind], value], KEY2CATDIM[kind])

    def forward(self, batch: Dict[str, Any], ucg_keys: Optional[List[str]] = None, set_ucg_rate_zero: bool = False,
                *args, **kwargs) -> Dict[str, Dict[str, torch.Tensor]]:
        forced = () if ucg_keys is None else tuple(ucg_keys)
        merged: Dict[str, torch.Tensor] = {}
        for conditioner in self.conditioners:
            drop = self._wants_drop(conditioner, forced, set_ucg_rate_zero)
            self._merge(merged, conditioner(batch, *args, force_zero_embedding=drop, **kwargs))
        return {"cond": merged}

    def on_fit_start(self, device: Optional[torch.device] = None, *args, **kwargs):
        for conditioner in self.conditioners:
            conditioner.on_fit_start(device, *args, **kwargs)

    def to(self, *args, **kwargs):
        moved = super().to(*args, **kwargs)
        for conditioner in moved.conditioners:      # children record their own placement attributes
            conditioner.to(*args, **kwargs)
        return moved

"""Checkpoint adaptation helpers with the behaviour of the reference's StateDictAdapter /
StateDictRenamer (src/lbm/trainer/utils.py:10-193): warm-starting a UNet whose conv_in /
to_k / to_v shapes differ from the checkpoint's by zero- or normal-padding, or narrowing."""
import logging
import re
import time
from typing import Dict, List, Literal, Optional

import torch


class StateDictAdapter:
    def _create_block(self, shape: List[int], strategy: str, input: Optional[torch.Tensor] = None):
        if strategy == "zeros":
            return torch.zeros(shape)
        if strategy == "normal":
            if input is not None:
                return torch.randn(shape) * input.std().item() + input.mean().item()
            return torch.randn(shape)
        raise ValueError(f"Unknown strategy {strategy}")

    def __call__(self, model_state_dict: Dict[str, torch.Tensor], checkpoint_state_dict: Dict[str, torch.Tensor],
                 regex_keys: Optional[List[str]] = None, strategy: Literal["zeros", "normal"] = "normal"):
        start = time.perf_counter()
        if regex_keys is None:
            regex_keys = list(model_state_dict.keys())
        patterns = [re.compile(r) for r in regex_keys]
        for key in list(checkpoint_state_dict.keys()):
            if not any(p.match(key) for p in patterns):
                continue
            dst_shape = model_state_dict[key].shape
            t = checkpoint_state_dict[key]
            if len(dst_shape) != t.dim():
                if t.dim() == 1:
                    t = t.unsqueeze(1)              # [a] -> [a, 1]
                elif len(dst_shape) == 1:
                    t = t[:, 0]                     # [a, b] -> [a]
                else:
                    raise ValueError(f"Shapes of {key} are different: {dst_shape} != {tuple(t.shape)}")
                assert len(dst_shape) == t.dim(), f"Shapes of {key} are different: {dst_shape} != {tuple(t.shape)}"
            if tuple(dst_shape) != tuple(t.shape):
                src_shape = tuple(t.shape)
                t = t.clone()
                for i, (d, s) in enumerate(zip(dst_shape, src_shape)):
                    if d > s:
                        missing = list(t.shape)
                        missing[i] = d - s
                        t = torch.cat((t, self._create_block(missing, strategy, t).to(t.dtype)), dim=i)
                    elif d < s:
                        t = t.narrow(i, 0, d)
                logging.info(f"Adapting {key} ({strategy}) from shape {src_shape} to {tuple(dst_shape)}")
            checkpoint_state_dict[key] = t
        logging.info(f"StateDictAdapter took {time.perf_counter() - start:.2f} seconds")
        return checkpoint_state_dict


class StateDictRenamer:
    """Moves exact keys: ``rename_dict`` maps an existing checkpoint key to its new name. A missing
    old key is skipped with a warning; a new name that already exists is an error."""

    def __call__(self, checkpoint_state_dict: Dict[str, torch.Tensor], rename_dict: Dict[str, str]):
        for old_key, new_key in rename_dict.items():
            if old_key not in checkpoint_state_dict:
                logging.warning(f"Key {old_key} not found in checkpoint state dict")
                continue
            assert new_key not in checkpoint_state_dict, f"Key {new_key} already exists in checkpoint state dict"
            checkpoint_state_dict[new_key] = checkpoint_state_dict.pop(old_key)
        return checkpoint_state_dict

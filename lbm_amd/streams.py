"""Two HIP streams for independent slices of a batch.

The VAE alternates MFMA-bound convolutions with HBM-bound GroupNorm passes.  Two halves of a batch on two streams
let one half's bandwidth-bound pass run beside the other half's convolution (the workgroup dispatcher mixes the two
queues), which a single in-order stream cannot do.  Results do not depend on the split: every kernel returns the same
bits for an image alone and inside a batch.  Inference only (autograd records on one stream).
``LBM_NO_STREAM_SPLIT=1`` keeps everything on the caller's stream (same-box A/B runs)."""
import os
from typing import Callable, Dict, List, Sequence, Tuple

import torch

_SPLIT = os.environ.get("LBM_NO_STREAM_SPLIT") is None
_side: Dict[int, Tuple[torch.cuda.Stream, torch.cuda.Stream]] = {}


def enabled(x: torch.Tensor) -> bool:
    return _SPLIT and x.is_cuda and not torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing()


def side_streams(device: torch.device) -> Tuple[torch.cuda.Stream, torch.cuda.Stream]:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _side:
        _side[idx] = (torch.cuda.Stream(device=idx), torch.cuda.Stream(device=idx))
    return _side[idx]


def run_slices(fn: Callable, parts: Sequence, device: torch.device = None) -> List[torch.Tensor]:
    """fn on every part, parts alternating between the two side streams; the caller's stream waits for both and may
    use the results (their memory is tied to it with record_stream, so the side pools do not recycle it early)."""
    dev = device if device is not None else parts[0].device
    cur = torch.cuda.current_stream(dev)
    streams = side_streams(dev)
    for s in streams:
        s.wait_stream(cur)
    outs = []
    for k, part in enumerate(parts):
        with torch.cuda.stream(streams[k % 2]):
            outs.append(fn(part))
    for s in streams:
        cur.wait_stream(s)
    for o in outs:
        o.record_stream(cur)
    return outs


def run_halves(fn: Callable[[torch.Tensor], torch.Tensor], x: torch.Tensor, min_half: int = 4) -> torch.Tensor:
    """fn(x) for a batch-wise fn, as two halves on two streams when the batch is large enough."""
    n = x.shape[0]
    if not enabled(x) or n < 2 * min_half:
        return fn(x)
    h = (n + 1) // 2
    a, b = run_slices(fn, (x[:h], x[h:]))
    return torch.cat((a, b), dim=0)

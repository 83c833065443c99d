"""The package's HIP streams: ONE set per device, created once, shared by everything that keeps batches in flight
(HamerEngine.contexts, the folder drivers of infer.py / d_infer.py).

HIP deals streams onto a few hardware queues in creation order.  Two streams on the same queue do not overlap at all:
of the 15 pairs among six streams three run two batches in flight at the serial rate (18.3 ms against 17.2-17.4 for the
others; tools/probes/stream_pairs.py, profiles/r03_stream_pairs.log), and which pair a component gets depends on how many
streams the process created before.  With one shared set the first n streams of the process are THE n streams, whoever asks."""
from typing import Dict, List

import torch

_STREAMS: Dict[int, List["torch.cuda.Stream"]] = {}


def get_streams(device, n: int) -> List["torch.cuda.Stream"]:
    """The first `n` of this device's shared streams (created on first use, in order)."""
    d = torch.device(device)
    key = d.index if d.index is not None else torch.cuda.current_device()      # "cuda" and "cuda:0" are the same device
    pool = _STREAMS.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]

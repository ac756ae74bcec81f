"""The two extra HIP streams of the training loop, one of each per device and process (shared by render() and the trainers)."""
from __future__ import annotations

import torch

_SIDE_STREAMS = {}


def side_stream(device) -> "torch.cuda.Stream":
    """ONE side stream per device for the whole process.  HIP multiplexes streams onto a handful of hardware queues; a
    process that keeps creating streams (a trainer per benchmark mode, say) sooner or later gets one that shares the
    current stream's queue, and work issued on it then serialises with the main chain instead of overlapping it
    (measured: the third trainer of a process ran 1.5x slower per step until its side stream was shared)."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _SIDE_STREAMS.get(key)
    if st is None:
        # ISR_SIDE_PRIORITY=-1: a high-priority side stream (A/B in DESIGN section 8: the chain's kernels then take their wave slots
        # ahead of the blend's instead of filling in behind them)
        import os
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev, priority=int(os.environ.get("ISR_SIDE_PRIORITY", "0")))
    return st


_EXTRA_SIDE = {}


def extra_side_stream(device, lane: int) -> "torch.cuda.Stream":
    """Further side streams (lane 1, 2, ...) for work that several binning chains can do next to each other."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(lane))
    st = _EXTRA_SIDE.get(key)
    if st is None:
        st = _EXTRA_SIDE[key] = torch.cuda.Stream(device=dev)
    return st


_MAIN_STREAMS = {}


def main_stream(device) -> "torch.cuda.Stream":
    """ONE high-priority stream per device (see SegTrainer.high_priority_main)."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _MAIN_STREAMS.get(key)
    if st is None:
        st = _MAIN_STREAMS[key] = torch.cuda.Stream(device=dev, priority=-1)
    return st

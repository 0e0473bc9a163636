"""Benchmark legs used by bench.py and the profiling scripts. They live outside the product package
because their CPU-baseline legs import the CPU checker (oracle/), which the product never does."""


_streams = {}


def two_streams(dev):
    """Two side streams per device, made once: a process has few hardware queues (four by default) and
    streams share them in the order they are made, so every measurement that needs two frames or batches
    in flight uses the same pair instead of making a pair of its own that may land on one queue."""
    import torch
    key = str(dev)
    if key not in _streams:
        _streams[key] = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    return _streams[key]

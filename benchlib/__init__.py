"""Benchmark legs used by bench.py and the profiling scripts. They live outside the product package
because their CPU-baseline legs import the CPU checker (oracle/), which the product never does."""

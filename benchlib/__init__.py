"""Workloads behind bench.py (one CLI, one JSON line per workload): common = scene, kernel attribution, rooflines, launch
plumbing; image = BASELINE configs[3]; relight = configs[4]; train = the training step and its gradient parity.  bench.py itself
holds the command line and the headline (batch) workload."""

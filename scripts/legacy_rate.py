#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in entry point: parallel_tempering_step called once per iteration
with host arrays (the unchanged F# host's call pattern), 10k nodes x 64 temperatures (diagnostic)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from procgen_amd import native as nt  # noqa: E402
from procgen_amd import workloads as wlm  # noqa: E402

n, T, calls = 10000, 64, 60
wl = wlm.graph_workload("c2", n, T, 1)
L = nt.load()
pos = wl["pos"][0].astype(np.float64).reshape(-1).copy()
ec = int(wl["ecount"][0, 0])
edges = np.ascontiguousarray(wl["edges"][0, 0, :ec]).reshape(-1)
costs = np.zeros(T)
temps = wl["temperatures"][0].copy()
for i in range(calls + 5):
    if i == 5:
        t0 = time.perf_counter()
    L.parallel_tempering_step(nt.ptr(pos), nt.ptr(edges), nt.ptr(costs), nt.ptr(temps), n, ec, T)
    assert L.pt_last_error() == 0, L.pt_last_error_string()
dt = (time.perf_counter() - t0) / calls
print(f"parallel_tempering_step: {dt * 1e3:.3f} ms per call (n={n}, T={T}, {pos.nbytes / 1e6:.1f} MB of positions each way) "
      f"= {T / dt:.0f} proposals/s PCIe-inclusive")

#!/usr/bin/env python3
"""10k-node graphs with more edges than the LDS-resident incremental kernel holds (diagnostic).
usage: dense_time.py [avg_degree] [flags]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from procgen_amd import workloads as wlm  # noqa: E402
from procgen_amd.engine import GraphEngine  # noqa: E402

deg = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n, R, iters = 10000, 64, 400
wl = wlm.graph_workload("dense", n, R, 1, avg_degree=deg)
eng = GraphEngine(n, wl["ecap"], R, 1, seed="dense", flags=flags)
eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
eng.run(1 + eng.swap_frequency)
eng.sync()
eng.kernel_time(reset=True)
eng.run(iters)
eng.sync()
ms, _ = eng.kernel_time()
print(f"n={n} R={R} ecap={wl['ecap']} info={eng.info()} us/step={ms * 1e3 / iters:.1f} proposals/s={R * iters / (ms * 1e-3):.3e} "
      f"status={eng.stats()['status_bits']}")
eng.close()

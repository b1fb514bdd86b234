#!/usr/bin/env python3
"""BASELINE configs[3] shape on one GPU: many independent 1k-node graphs x 16 temperatures (diagnostic).
usage: config4_time.py [graphs] [iters] [flags]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from procgen_amd import workloads as wlm  # noqa: E402
from procgen_amd.engine import GraphEngine  # noqa: E402

a = sys.argv[1:]
G = int(a[0]) if len(a) > 0 else 512
iters = int(a[1]) if len(a) > 1 else 252
flags = int(a[2]) if len(a) > 2 else 0
n, T = 1000, 16
distinct = min(G, 16)
wl = wlm.graph_workload("c4", n, T, distinct)
rep = (G + distinct - 1) // distinct
tile = lambda x: np.tile(x, (rep,) + (1,) * (x.ndim - 1))[:G]  # noqa: E731
eng = GraphEngine(n, wl["ecap"], T, G, seed="procgen-bench/c4", flags=flags, block_threads=int(os.environ.get("BLOCK", "0")))
eng.upload(tile(wl["pos"]), tile(wl["edges"]), tile(wl["ecount"]), tile(wl["temperatures"]), tile(wl["kind"]), tile(wl["value"]))
eng.run(1 + eng.swap_frequency)
eng.sync()
eng.kernel_time(reset=True)
eng.run(iters)
eng.sync()
ms, launches = eng.kernel_time()
st = eng.stats()
res = eng.download(positions=False, edges=False)
terms = eng.cost_terms()
assert st["status_bits"] == 0 and np.array_equal(terms[:, :, 6], res["cost"])
print(f"graphs={G} n={n} T={T} info={eng.info()} kernel_ms={ms:.3f} us/step={ms * 1e3 / iters:.2f} "
      f"proposals/s={G * T * iters / (ms * 1e-3):.3e} accept={st['accepted'] / st['proposals']:.3f}")
eng.close()

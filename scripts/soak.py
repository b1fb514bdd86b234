#!/usr/bin/env python3
"""Long runs of the graph engine: no status bit may come up and the tracked cost must equal a fresh
evaluation of the final state (diagnostic).  usage: soak.py [n] [replicas] [steps] [flags] [cluster]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from procgen_amd import workloads as wlm  # noqa: E402
from procgen_amd.engine import GraphEngine  # noqa: E402

a = sys.argv[1:]
n = int(a[0]) if len(a) > 0 else 10000
R = int(a[1]) if len(a) > 1 else 64
steps = int(a[2]) if len(a) > 2 else 400000
flags = int(a[3]) if len(a) > 3 else 0
cluster = int(a[4]) if len(a) > 4 else 0
wl = wlm.graph_workload("soak", n, R, 1)
eng = GraphEngine(n, wl["ecap"], R, 1, seed="soak", flags=flags, cluster_size=cluster)
eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
t0 = time.time()
done = 0
while done < steps:
    k = min(50000, steps - done)
    eng.run(k)
    eng.sync()
    done += k
    st = eng.stats()
    print(f"  {done} steps, {time.time() - t0:.1f} s, accepted {st['accepted']}, swaps {st['swaps']}, status {st['status_bits']:#x}", flush=True)
    assert st["status_bits"] == 0
res = eng.download(positions=False)
terms = eng.cost_terms()
assert np.array_equal(terms[:, :, 6], res["cost"]), "tracked cost differs from a fresh evaluation"
print(f"n={n} R={R} steps={steps} info={eng.info()}: ok, edges {res['ecount'].min()}..{res['ecount'].max()}, "
      f"reach {terms[:, :, 7].min():.0f}..{terms[:, :, 7].max():.0f}")
eng.close()

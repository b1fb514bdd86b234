#!/usr/bin/env python3
"""Same job through the global-memory incremental kernel with 1 and with several workgroups per replica:
final costs, edge counts and accept counts must agree exactly (diagnostic).
usage: cluster_check.py [n] [replicas] [iters] [members]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from procgen_amd import native as nt  # noqa: E402
from procgen_amd import workloads as wlm  # noqa: E402
from procgen_amd.engine import GraphEngine  # noqa: E402

a = sys.argv[1:]
n = int(a[0]) if len(a) > 0 else 300000
R = int(a[1]) if len(a) > 1 else 16
iters = int(a[2]) if len(a) > 2 else 300
members = int(a[3]) if len(a) > 3 else 16
wl = wlm.graph_workload("cc", n, R, 1)
out = []
for cs in (1, members):
    eng = GraphEngine(n, wl["ecap"], R, 1, seed="cluster-check", flags=nt.FLAG_STREAMING_INCREMENTAL, cluster_size=cs,
                      swap_frequency=50)
    eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
    eng.run(iters)
    eng.sync()
    res = eng.download()
    st = eng.stats()
    terms = eng.cost_terms()
    assert st["status_bits"] == 0, st
    assert np.array_equal(terms[:, :, 6], res["cost"]), "tracked cost differs from a fresh evaluation"
    out.append((res, st, eng.info()["cluster_size"], eng.kernel_time()[0]))
    eng.close()
(r1, s1, c1, t1), (r2, s2, c2, t2) = out
print(f"n={n} R={R} iters={iters}: members {c1} -> {t1:.1f} ms, members {c2} -> {t2:.1f} ms; accepted {s1['accepted']} / {s2['accepted']}")
assert s1["accepted"] == s2["accepted"] and s1["swaps"] == s2["swaps"]
for k in ("cost", "ecount", "pos", "edges"):
    assert np.array_equal(r1[k], r2[k]), k
print("identical")

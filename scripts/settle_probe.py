#!/usr/bin/env python3
"""Is the rise of the headline rate over the first second of a run the chip's clock settling, or the chains' state
changing (edge lists shrink at the hot end of the ladder)?  Times 20 swap intervals (a) from a cold start, (b) after 600
intervals of load, (c) after the same state has sat idle for 3 s, (d) right after (c).  State moves on from (a) to (d);
the clock is cold at (a) and (c) only.  (Measured: (c) equals (b) -- it is the state.)  usage: settle_probe.py [chunks]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from procgen_amd import workloads as wlm  # noqa: E402
from procgen_amd.engine import ChunkEngine, GraphEngine  # noqa: E402

chunks = len(sys.argv) > 1 and sys.argv[1] == "chunks"
if chunks:
    G, T = 4096, 6
    wl = wlm.chunk_workload("c3", 16, T)
    tile = lambda a: np.tile(a, (G // 16,) + (1,) * (a.ndim - 1))  # noqa: E731
    eng = ChunkEngine(T, G, seed="procgen-bench/c3")
    eng.upload(tile(wl["density"]), tile(wl["moisture"]), tile(wl["temperature"]), tile(wl["material"]), tile(wl["temperatures"]))
    per = G * T
else:
    n, R = 10000, 64
    wl = wlm.graph_workload("c2", n, R, 1)
    eng = GraphEngine(n, wl["ecap"], R, 1, seed="procgen-bench/c2")
    eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
    per = R
f = eng.swap_frequency
eng.run(1)
eng.sync()


def timed(tag, k=20):
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(k):
        eng.run(f)
    eng.sync()
    dt = time.perf_counter() - t0
    print(f"{tag:34s} {per * f * k / dt:.4e} proposals/s", flush=True)


time.sleep(3.0)
timed("(a) cold start")
for _ in range(600):
    eng.run(f)
timed("(b) after 600 intervals of load")
time.sleep(3.0)
timed("(c) same state, after 3 s idle")
timed("(d) right after (c)")
for _ in range(600):
    eng.run(f)
timed("(e) after 600 more intervals")
eng.close()

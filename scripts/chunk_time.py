#!/usr/bin/env python3
"""Kernel time per proposal of the voxel-chunk engine (diagnostic).
usage: chunk_time.py [chunks] [iters] [refinement_probability]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from procgen_amd import workloads as wlm  # noqa: E402
from procgen_amd.engine import ChunkEngine  # noqa: E402

a = sys.argv[1:]
G = int(a[0]) if len(a) > 0 else 4096
iters = int(a[1]) if len(a) > 1 else 500
rp = float(a[2]) if len(a) > 2 else None
T = 6
distinct = min(G, 32)
wl = wlm.chunk_workload("c3", distinct, T)
rep = (G + distinct - 1) // distinct
tile = lambda x: np.tile(x, (rep,) + (1,) * (x.ndim - 1))[:G]  # noqa: E731
eng = ChunkEngine(T, G, seed="procgen-bench/c3", refinement_probability=rp)
eng.upload(tile(wl["density"]), tile(wl["moisture"]), tile(wl["temperature"]), tile(wl["material"]), tile(wl["temperatures"]))
eng.run(1 + eng.swap_frequency)
eng.sync()
eng.kernel_time(reset=True)
eng.run(iters)
eng.sync()
ms, launches = eng.kernel_time()
st = eng.stats()
print(f"chunks={G} T={T} refine_p={rp} kernel_ms={ms:.3f} us/step={ms * 1e3 / iters:.2f} proposals/s={G * T * iters / (ms * 1e-3):.3e} "
      f"accept={st['accepted'] / st['proposals']:.3f}")
eng.close()

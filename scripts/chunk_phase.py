#!/usr/bin/env python3
"""Per-phase tick shares of the chunk step kernel from the diagnostic stamps build (lane 0 of every wave).
usage: chunk_phase.py [chunks] [iters]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PROCGEN_PT_LIB", os.path.join(ROOT, "procgen_amd", "lib_stamps", "libgpu_compute.so"))
from procgen_amd import native as nt  # noqa: E402
from procgen_amd import workloads as wlm  # noqa: E402
from procgen_amd.engine import ChunkEngine  # noqa: E402

NAMES = ["loop top / cone commit", "draws", "decode + voxel loads issued", "geological", "overhang, erosion, counts", "biome, aesthetics",
         "cave search", "cost + accept", "commit"]
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 500
T = 6
L = nt.load()
distinct = min(G, 32)
wl = wlm.chunk_workload("c3", distinct, T)
rep = (G + distinct - 1) // distinct
tile = lambda x: np.tile(x, (rep,) + (1,) * (x.ndim - 1))[:G]  # noqa: E731
eng = ChunkEngine(T, G, seed="procgen-bench/c3")
eng.upload(tile(wl["density"]), tile(wl["moisture"]), tile(wl["temperature"]), tile(wl["material"]), tile(wl["temperatures"]))
eng.run(1 + eng.swap_frequency)
eng.sync()
buf = np.zeros((4096, 16), np.uint64)
L.pt_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), 1)
eng.kernel_time(reset=True)
eng.run(iters)
eng.sync()
ms, _ = eng.kernel_time()
L.pt_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), 1)
st = buf[buf.sum(axis=1) > 0].astype(np.float64)
tot = st.sum(axis=1)
print(f"chunks={G} iters={iters} kernel us/step={ms * 1e3 / iters:.2f} waves stamped={len(st)} ticks/step mean {tot.mean() / iters:.0f}")
for k, name in enumerate(NAMES):
    print(f"  {name:32s} {st[:, k].mean() / iters:8.0f} ticks  {100 * st[:, k].sum() / tot.sum():5.1f}%")
eng.close()

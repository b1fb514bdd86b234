#!/usr/bin/env python3
"""Per-phase tick shares of the clustered streaming kernel (pt_graph_bigc_kernel) from the diagnostic stamps build:
the leader workgroup's first thread, one row per replica.  usage: bigc_phase.py [nodes] [replicas] [iters] [tmin] [tmax]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PROCGEN_PT_LIB", os.path.join(ROOT, "procgen_amd", "lib_stamps", "libgpu_compute.so"))
from procgen_amd import native as nt  # noqa: E402
from procgen_amd import workloads as wlm  # noqa: E402
from procgen_amd.engine import GraphEngine  # noqa: E402

NAMES = ["loop top", "draws, decode, duplicate / count scan", "perturbation: pass 1 + exchange", "perturbation: pass 2 + exchange",
         "removal analysis scan", "subtree: init", "subtree: pointer-jumping rounds", "subtree: bitmap", "sweeps",
         "(other topology work) to decision", "decision", "apply", "closing barrier"]
a = sys.argv[1:]
n = int(a[0]) if len(a) > 0 else 1000000
R = int(a[1]) if len(a) > 1 else 16
iters = int(a[2]) if len(a) > 2 else 300
tmin = float(a[3]) if len(a) > 3 else 0.5
tmax = float(a[4]) if len(a) > 4 else 5.0
L = nt.load()
wl = wlm.graph_workload("c2", n, R, 1, t_min=tmin, t_max=tmax)
eng = GraphEngine(n, wl["ecap"], R, 1, seed="procgen-bench/c2")
eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
eng.run(1 + eng.swap_frequency if eng.swap_frequency < 200 else 1)
eng.sync()
buf = np.zeros((4096, 16), np.uint64)
L.pt_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), 1)
eng.kernel_time(reset=True)
eng.run(iters)
eng.sync()
ms, _ = eng.kernel_time()
L.pt_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), 1)
st = buf[:R].astype(np.float64)
tot = st.sum(axis=1)
print(f"n={n} replicas={R} iters={iters} T=[{tmin},{tmax}] info={eng.info()} kernel us/step={ms * 1e3 / iters:.1f}")
print("ticks per step by replica:", np.round(tot / iters).astype(int).tolist())
for k, name in enumerate(NAMES):
    print(f"  {name:42s} mean {st[:, k].mean() / iters:9.0f}  {100 * st[:, k].sum() / tot.sum():5.1f}%   slowest replica {st[np.argmax(tot), k] / iters:9.0f}")
eng.close()

#!/usr/bin/env python3
"""Ticks per step of the incremental graph kernel by kind of step and temperature slot, from a build with -DPT_STAMPS_LIGHT
(make -C procgen_amd/csrc OUT_DIR=../lib_light EXTRA=-DPT_STAMPS_LIGHT; two clock reads per step, nothing else differs
from the shipped kernel).  usage: kind_profile.py [nodes] [replicas] [iters] [tmin] [tmax]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PROCGEN_PT_LIB", os.path.join(ROOT, "procgen_amd", "lib_light", "libgpu_compute.so"))
from procgen_amd import native as nt  # noqa: E402
from procgen_amd import workloads as wlm  # noqa: E402
from procgen_amd.engine import GraphEngine  # noqa: E402

a = sys.argv[1:]
n = int(a[0]) if len(a) > 0 else 10000
R = int(a[1]) if len(a) > 1 else 64
iters = int(a[2]) if len(a) > 2 else 2000
tmin = float(a[3]) if len(a) > 3 else 0.5
tmax = float(a[4]) if len(a) > 4 else 5.0
L = nt.load()
wl = wlm.graph_workload("c2", n, R, 1, t_min=tmin, t_max=tmax)
eng = GraphEngine(n, wl["ecap"], R, 1, seed="procgen-bench/c2")
eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
eng.run(1 + eng.swap_frequency)
eng.sync()
buf = np.zeros((4096, 16), np.uint64)
L.pt_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), 1)
eng.kernel_time(reset=True)
eng.run(iters)
eng.sync()
ms, launches = eng.kernel_time()
L.pt_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), 1)
st = buf[:R].astype(np.float64)
slots = eng.replica_slots().reshape(-1)
order = np.argsort(slots)
ticks, counts = st[order, 0:6], st[order, 8:14]
tot = ticks.sum(axis=1)
print(f"n={n} R={R} iters={iters} T=[{tmin},{tmax}] kernel us/step={ms * 1e3 / iters:.2f}; ticks per ns over the slowest replica: {tot.max() / (ms * 1e6):.3f}")
print("ticks per step by temperature slot, hottest first (the slot a replica holds at the end):")
print("  ", np.round(tot / iters).astype(int).tolist())
names = ["perturbation", "addition", "removal", "removal + listed search", "nothing to do", "removal + general search"]
for half, sel in (("slots above 2.0", wl["temperatures"][0][slots[order]] > 2.0), ("slots at or below 2.0", wl["temperatures"][0][slots[order]] <= 2.0)):
    if not sel.any():
        continue
    print(f"{half}: {int(sel.sum())} replicas, {tot[sel].mean() / iters:.0f} ticks per step on average, {tot[sel].max() / iters:.0f} the slowest")
    for k, nm in enumerate(names):
        c, t = counts[sel, k].sum(), ticks[sel, k].sum()
        if c:
            print(f"   {nm:24s} {100 * c / counts[sel].sum():5.1f} % of the steps, {t / c:8.0f} ticks each, {100 * t / ticks[sel].sum():5.1f} % of the time")
w = st[:, 7].sum() / max(st[:, 15].sum(), 1.0)
calls = max(st[:, 15].sum(), 1.0)
print(f"the walk through the lists (delta_list_search): {int(st[:, 15].sum())} calls, {w:.0f} ticks each, "
      f"{st[:, 6].sum() / calls:.1f} rounds down the tree and {st[:, 14].sum() / calls:.1f} rounds of label propagation per call")
eng.close()

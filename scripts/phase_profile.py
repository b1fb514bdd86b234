#!/usr/bin/env python3
"""Per-phase cycle shares of the graph step kernel, from the diagnostic stamps build
(make -C procgen_amd/csrc stamps; PROCGEN_PT_LIB=procgen_amd/lib_stamps/libgpu_compute.so).
Shares only: the stamped build is never the one benchmarked."""
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

NAMES = ["zero", "nodes+edges", "scan", "fill", "collide+dist1", "dist2", "bfs", "combine", "draw+decode", "accept",
         "apply", "coll:loops", "coll:price", "coll:reduce", "-", "-"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    block = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    L = nt.load()
    wl = wlm.graph_workload("c2", n, R, 1)
    eng = GraphEngine(n, wl["ecap"], R, 1, seed="procgen-bench/c2", block_threads=block)
    eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
    eng.run(1 + eng.swap_frequency)
    eng.sync()
    buf = np.zeros((4096, 16), np.uint64)
    L.pt_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), 1)
    eng.run(iters)
    eng.sync()
    ms, launches = eng.kernel_time()
    L.pt_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), 1)
    st = buf[:R].astype(np.float64)
    tot = st.sum(axis=1)
    print(f"n={n} replicas={R} iters={iters} kernel_ms={ms:.2f} us/proposal(wall)={ms * 1e3 / iters:.1f}")
    print(f"cycles/proposal: mean {tot.mean() / iters:.0f} max {tot.max() / iters:.0f} (slowest replica {tot.argmax()})")
    for k, name in enumerate(NAMES):
        if st[:, k].sum() > 0:
            print(f"  {name:14s} mean {st[:, k].mean() / iters:9.0f} cyc  {100 * st[:, k].sum() / tot.sum():5.1f}%   "
                  f"hot(r0) {st[0, k] / iters:9.0f}  cold(r{R-1}) {st[R-1, k] / iters:9.0f}")
    eng.close()


if __name__ == "__main__":
    main()

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
         "apply", "coll:loops", "coll:price", "coll:reduce", "handoff wait", "-"]
# the incremental kernel stamps the same slots with its own phases
NAMES_INC = ["centroid", "pairs+dist1+edges", "subtree:bitmap", "sweeps", "search 2 / end", "dist2", "edge term", "subtree:init",
             "decode", "decide", "apply", "edge loads issued", "next draws", "subtree:rounds", "one stamp", "(searches)"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    block = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    cluster = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    L = nt.load()
    ponly = int(os.environ.get("PERTURB_ONLY", "0"))
    wl = wlm.graph_workload("c2", n, R, 1)
    flags = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    if os.environ.get("ALL_TEMPS"):                 # every replica at one temperature (5.0: four topology moves in five)
        wl["temperatures"][:] = float(os.environ["ALL_TEMPS"])
    eng = GraphEngine(n, wl["ecap"], R, 1, seed="procgen-bench/c2", block_threads=block, cluster_size=cluster, flags=flags,
                      mutation_mode=nt.MUT_PERTURB_ONLY if ponly else nt.MUT_ADAPTIVE)
    inc = eng.info()["incremental"]
    eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
    eng.run(1 + eng.swap_frequency)
    eng.sync()
    buf = np.zeros((4096, 16), np.uint64)
    L.pt_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), 1)
    eng.run(iters)
    eng.sync()
    ms, launches = eng.kernel_time()
    L.pt_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), 1)
    CS = 1 if inc else (cluster if cluster else 4)
    nblk = R if CS == 1 else (R + 7) // 8 * 8 * CS
    st_all = buf[:nblk].astype(np.float64)
    print(f"n={n} replicas={R} iters={iters} cluster={CS} kernel_ms={ms:.2f} us/proposal(wall)={ms * 1e3 / iters:.1f}")
    roles = {"all": np.arange(nblk)} if CS == 1 else {
        "leader(pair)": np.array([b for b in range(nblk) if (b >> 3) % CS == 0]),
        "pair": np.array([b for b in range(nblk) if 0 < (b >> 3) % CS < CS - 1]),
        "bfs+dist": np.array([b for b in range(nblk) if (b >> 3) % CS == CS - 1])}
    for role, ids in roles.items():
        if len(ids) == 0:
            continue
        st = st_all[ids]
        tot = st.sum(axis=1)
        print(f"[{role}] cycles/proposal: mean {tot.mean() / iters:.0f} max {tot.max() / iters:.0f}")
        if CS == 1:
            slots = eng.replica_slots().reshape(-1)             # temperature slot of every replica (slot 0 = hottest)
            order = np.argsort(slots[:len(tot)])
            print("  cycles/proposal by temperature slot, hottest first:", np.round(tot[order] / iters).astype(int).tolist())
        for k, name in enumerate(NAMES_INC if inc else NAMES):
            if st[:, k].sum() > 0:
                if name.startswith("("):
                    print(f"  {name:18s} per proposal: mean {st[:, k].mean() / iters:.4f} max {st[:, k].max() / iters:.4f}")
                    continue
                print(f"  {name:18s} mean {st[:, k].mean() / iters:9.0f} cyc  {100 * st[:, k].sum() / tot.sum():5.1f}%"
                      f"   slowest replica {st[np.argmax(tot), k] / iters:9.0f}")
    if buf[2048:2048 + 64].sum() > 0:
        print('per-wave cycles (mean over replicas), fine stamps build:')
        print('  pass 1, from its start to its sums added:', np.round(buf[3072:3072 + R].astype(np.float64).mean(axis=0) / iters).astype(int).tolist())
        for m, nm in enumerate(["pass 1: before the loop", "pass 1: the loop", "pass 1: near nodes listed", "pass 1: sums added",
                                "pass 2: before the loop", "pass 2: the loop", "pass 2: sums added"]):
            rows = buf[1024 + 64 * m:1024 + 64 * m + min(R, 64)].astype(np.float64)
            if rows.sum() > 0:
                print(f'    {nm:28s}', np.round(rows.mean(axis=0) / iters).astype(int).tolist())
        print('  pass 2, from the barrier to its sums added:', np.round(buf[2048:2048 + R].astype(np.float64).mean(axis=0) / iters).astype(int).tolist())
    eng.close()


if __name__ == "__main__":
    main()

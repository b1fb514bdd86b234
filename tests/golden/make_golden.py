#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/pt_oracle.c, canonical-sum mode).

The reference cannot be run in this environment (F#/.NET), and it records no trajectories of
its own, so these vectors pin the ORACLE's behaviour (and through it the GPU's) against
accidental change; they are not outputs of the reference.  "parity unpinned" beyond the
reference's scalar known-answer values (tests/test_oracle_kat.py).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_binding as ob  # noqa: E402
import parity_util as pu  # noqa: E402
from procgen_amd import workloads as wlm  # noqa: E402

CASES = {
    # BASELINE configs[0]: 20 nodes, 5 temperatures 0.5-5.0, 1000 iterations, sphere start without edges
    "config1": dict(n=20, T=5, iters=1000, f=8, layout="sphere", empty=True, seed="config1"),
    # a small graph with edges: topology moves change the reached set
    "graph120": dict(n=120, T=4, iters=400, f=21, layout="box", empty=False, seed="golden-graph120"),
}


def case_workload(c):
    wl = wlm.graph_workload("golden-" + c["seed"], c["n"], c["T"], 1, layout=c["layout"], edge_headroom=3.0)
    if c["empty"]:
        wl["ecount"][:] = 0
    return wl


CHUNK_CASE = dict(chunks=1, T=3, iters=300, f=11, seed="golden-chunk")


def chunk_workload():
    return wlm.chunk_workload("golden", CHUNK_CASE["chunks"], CHUNK_CASE["T"])


def make_chunk():
    """Voxel-chunk run (terrain oracle): the trace and the final costs only (the fields are 384 KB per replica)."""
    import terrain_binding as tb
    c = CHUNK_CASE
    wl = chunk_workload()
    o = tb.run(wl["density"][0], wl["moisture"][0], wl["temperature"][0], wl["material"][0], wl["temperatures"][0],
               c["iters"], c["f"], tb.default_config(ob.SUM_CANONICAL), key=int(ob.lib().og_seed_key(c["seed"].encode())))
    assert o["err"] == 0
    tr = o["trace"]
    np.savez_compressed(os.path.join(HERE, "chunk.npz"), kind=tr["kind"].astype(np.int8), a=tr["a"].astype(np.int32),
                        b=tr["b"].astype(np.int8), flags=tr["flags"].astype(np.uint8), reach0=tr["reach0"].astype(np.int32),
                        reach_origin=tr["reach_origin"].astype(np.int32), cost_after=tr["cost_after"],
                        swap_trace=o["swap_trace"], cost=o["cost"],
                        density_sum=o["density"].astype(np.float64).sum(axis=1))
    print("chunk accepted", int(((tr["flags"] & 4) != 0).sum()), "of", tr["flags"].size, "swaps", int(o["swap_trace"].sum()))


def main():
    make_chunk()
    for name, c in CASES.items():
        wl = case_workload(c)
        o = pu.oracle_run(wl, c["iters"], c["f"], seed=c["seed"])[0]
        assert o["err"] == 0
        tr = o["trace"]
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            kind=tr["kind"].astype(np.int8), a=tr["a"].astype(np.int32), b=tr["b"].astype(np.int32),
            flags=tr["flags"].astype(np.uint8), reach0=tr["reach0"].astype(np.int32),
            reach_origin=tr["reach_origin"].astype(np.int32), cost_after=tr["cost_after"],
            swap_trace=o["swap_trace"], pos=o["pos"], cost=o["cost"], ecount=o["ecount"],
            edges=o["edges"][:, :int(o["ecount"].max())])
        print(name, "accepted", int(((tr["flags"] & 4) != 0).sum()), "of", tr["flags"].size, "swaps", int(o["swap_trace"].sum()))


if __name__ == "__main__":
    main()

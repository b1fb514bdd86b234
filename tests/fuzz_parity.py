"""Randomised parity sweeps of the HIP engines against the CPU oracle (GPU needed).

Not collected by pytest: run it by hand, `python tests/fuzz_parity.py [cases] [seed] [graph|chunk|collide|large|traceoff|cluster]`.
  graph    every case draws a graph shape, a layout scale, node roles, a temperature ladder, a move set, cost
           weights, the draws (Philox, or injected and salted with extreme values), a swap interval, a launch
           split and one of the engine's kernel routes;
  large    the same for 14k to 110k nodes through the global-memory routes;
  chunk    the voxel-chunk engine: air share, opened boxes, uniform chunks, fractional densities, ladders, weights;
  traceoff sparse graphs on hot ladders with NO step trace recorded (the production paths of the incremental
           kernels, early rejection of tree-edge removals included): final state, swaps, slot map, counts;
  cluster  the graph cases through the clustered streaming kernel only (2 to 16 workgroups per replica), with node
           degrees on both sides of what its per-node lists of incident edges hold, uploaded self loops, hot ladders,
           half of the cases without a step trace;
  collide  collision_detection, both entries: clumps, coincident points, points planted at the bounds, far offsets.
Each case runs on the GPU through the C ABI and on the oracle, and the results (for the engines: the whole
trace) are compared bit for bit.  A failing case prints the numbers needed to repeat it.
"""
import os
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import oracle_binding as ob          # noqa: E402
import parity_util as pu             # noqa: E402
from procgen_amd import native as nt      # noqa: E402
from procgen_amd import workloads as wlm  # noqa: E402

ROUTES = [
    ("incremental", dict(flags=0)),
    ("incremental-512", dict(flags=0, block_threads=512)),
    ("incremental-1024", dict(flags=0, block_threads=1024)),
    ("full-lds", dict(flags=nt.FLAG_FULL_EVALUATION)),
    ("full-global", dict(flags=nt.FLAG_GLOBAL_MEMORY_PLACEMENT)),
    ("incremental-global", dict(flags=nt.FLAG_STREAMING_INCREMENTAL)),
    ("incremental-global-c3", dict(flags=nt.FLAG_STREAMING_INCREMENTAL, cluster_size=3)),
    ("full-cluster2", dict(flags=nt.FLAG_FULL_EVALUATION, cluster_size=2)),
    ("full-cluster4", dict(flags=nt.FLAG_FULL_EVALUATION, cluster_size=4)),
]


LARGE = False                  # graphs beyond what the LDS-resident kernels hold: the global-memory routes
LARGE_ROUTES = [
    ("auto", dict(flags=0)),
    ("incremental-global", dict(flags=nt.FLAG_STREAMING_INCREMENTAL)),
    ("incremental-global-c2", dict(flags=nt.FLAG_STREAMING_INCREMENTAL, cluster_size=2)),
    ("incremental-global-c5", dict(flags=nt.FLAG_STREAMING_INCREMENTAL, cluster_size=5)),
    ("incremental-global-c16", dict(flags=nt.FLAG_STREAMING_INCREMENTAL, cluster_size=16)),
    ("full-global", dict(flags=nt.FLAG_GLOBAL_MEMORY_PLACEMENT)),
]


def make_case(rng, index):
    n = int(rng.choice([1, 2, 3, 5, 17, 64, 65, int(rng.integers(6, 400)), int(rng.integers(400, 1500))]))
    T = int(rng.integers(1, 9))
    G = int(rng.integers(1, 4))
    iters = int(rng.integers(20, 350))
    if LARGE:
        n, T, G, iters = int(rng.integers(14000, 110000)), int(rng.integers(1, 4)), 1, int(rng.integers(20, 70))
    f = int(rng.choice([1, 2, int(rng.integers(3, 60)), iters + 5]))
    deg = float(rng.choice([2.0, 2.1, 3.0, 4.0, 8.0]))
    n_edges = max(int(n * deg / 2.0), n - 1)
    headroom = 1.0 + (iters + 8) / max(n_edges, 1)
    layout, scale = str(rng.choice(["box", "sphere"])), float(rng.choice([0.2, 0.6, 1.0, 1.7, 4.0]))
    if LARGE:                                       # keep the number of colliding pairs linear in n
        layout, scale = "box", max(scale, 0.6)
    wl = wlm.graph_workload("fuzz", n, T, G, layout=layout, avg_degree=deg, edge_headroom=headroom, index=index)
    # layout scale: from everything colliding to nothing colliding
    wl["pos"] *= np.float32(scale)
    # edge list surgery: drop a random share (fragments the graph), then maybe repeat a few
    for g in range(G):
        ec = int(wl["ecount"][g, 0])
        e = wl["edges"][g, 0, :ec].copy()
        mode = rng.integers(0, 4)
        if mode == 1 and ec:
            e = e[rng.random(ec) > rng.choice([0.1, 0.5, 0.9])]
        elif mode == 2:
            e = e[:0]
        elif mode == 3 and ec > 4:
            k = int(rng.integers(1, min(ec, 12)))
            dup = e[rng.integers(0, ec, size=k)]
            if rng.random() < 0.5:
                dup = dup[:, ::-1]
            e = np.concatenate([e, dup])
        if len(e) > wl["ecap"] - iters - 1:
            e = e[:wl["ecap"] - iters - 1]
        wl["edges"][g, :, :] = 0
        wl["edges"][g, :, :len(e)] = e
        wl["ecount"][g, :] = len(e)
        # node roles: sometimes move the two endpoints, sometimes drop one of them
        if n >= 3 and rng.random() < 0.5:
            kind = np.full(n, nt.NODE_STANDARD, np.uint8)
            a, b = rng.choice(n, 2, replace=False)
            r = rng.random()
            if r < 0.8:
                kind[a] = nt.NODE_ORIGIN
            if r > 0.1:
                kind[b] = nt.NODE_TERMINAL
            wl["kind"][g] = kind
    # temperatures: the usual ladder, an unsorted one, or one with a very cold replica
    tm = rng.integers(0, 3)
    if tm == 1:
        wl["temperatures"][:] = rng.uniform(0.05, 12.0, size=(G, T))
    elif tm == 2:
        wl["temperatures"][:, 0] = 1e-3
    mutation_mode = int(rng.choice([nt.MUT_ADAPTIVE, nt.MUT_ADAPTIVE, nt.MUT_PERTURB_ONLY]))
    weights = None if rng.random() < 0.5 else [float(x) for x in rng.uniform(0.0, 2.5, size=6)]
    cut = int(rng.integers(0, iters + 1))
    chunks = [c for c in (cut, iters - cut) if c > 0]
    route = ROUTES[int(rng.integers(0, len(ROUTES)))]
    if LARGE:
        route = LARGE_ROUTES[int(rng.integers(0, len(LARGE_ROUTES)))]
    draws = None
    if T >= 2 and rng.random() < 0.2:               # injected draws salted with the extreme values
        passes = (iters + f - 1) // f
        cu = rng.random((G, T, iters, nt.DRAWS_PER_STEP))
        su = rng.random((G, passes, T - 1))
        for arr in (cu, su):
            salt = rng.random(arr.shape)
            arr[salt < 0.02] = 0.0
            arr[(salt >= 0.02) & (salt < 0.04)] = np.nextafter(1.0, 0.0)
            arr[(salt >= 0.04) & (salt < 0.05)] = 5e-324
            arr[(salt >= 0.05) & (salt < 0.06)] = 0.5
        draws = (cu, su)
    return dict(wl=wl, iters=iters, f=f, mutation_mode=mutation_mode, weights=weights, chunks=chunks, route=route,
                draws=draws,
                summary=f"n={n} T={T} G={G} iters={iters} f={f} deg={deg} E={wl['ecount'][:, 0].tolist()} "
                        f"mut={mutation_mode} w={'custom' if weights else 'default'} temps={tm} chunks={chunks} "
                        f"route={route[0]} draws={'injected' if draws else 'philox'}")


CLUSTER = False                # the clustered streaming kernel only
CLUSTER_ROUTES = [(f"incremental-global-c{c}", dict(flags=nt.FLAG_STREAMING_INCREMENTAL, cluster_size=c)) for c in (2, 3, 5, 7, 16)]


_make_graph_case = make_case


def make_cluster_case(rng, index):
    case = _make_graph_case(rng, index)
    wl = case["wl"]
    n, G, T = wl["n"], wl["groups"], wl["temps"]
    iters = case["iters"]
    if n >= 8 and rng.random() < 0.6:
        # a denser list: mean degree 10 to 24, so that nodes sit on both sides of 16 incident edges
        deg = float(rng.choice([10.0, 15.0, 17.0, 24.0]))
        n_edges = min(max(int(n * deg / 2.0), n - 1), n * (n - 1) // 2)
        dense = wlm.graph_workload("fuzz-dense", n, T, G, avg_degree=deg, edge_headroom=1.0 + (iters + 16) / max(n_edges, 1), index=index)
        dense["pos"], dense["kind"], dense["value"], dense["temperatures"] = wl["pos"], wl["kind"], wl["value"], wl["temperatures"]
        wl = dense
    if rng.random() < 0.3:                         # self loops in the uploaded list
        for g in range(G):
            ec = int(wl["ecount"][g, 0])
            k = int(min(rng.integers(1, 6), wl["ecap"] - iters - 1 - ec))
            if k > 0:
                v = rng.integers(0, n, size=k).astype(np.int32)
                wl["edges"][g, :, ec:ec + k] = np.stack([v, v], axis=1)
                wl["ecount"][g, :] = ec + k
    if rng.random() < 0.5 and T > 1:               # hot: four proposals out of five add or remove an edge
        wl["temperatures"][:] = nt.geometric_ladder(float(rng.choice([0.6, 2.5])), float(rng.choice([6.0, 12.0])), T)
    case["wl"] = wl
    # (a clustered grid must be co-resident: replicas rounded up to eight, times the members, within the 256 CUs)
    fits = [rt for rt in CLUSTER_ROUTES if (G * T + 7) // 8 * 8 * rt[1]["cluster_size"] <= 256]
    case["route"] = fits[int(rng.integers(0, len(fits)))]
    case["trace"] = bool(rng.random() < 0.5)
    case["summary"] += f" -> cluster route={case['route'][0]} E={wl['ecount'][:, 0].tolist()} trace={case['trace']}"
    return case


def run_case(case, seed):
    wl = case["wl"]
    omut = {nt.MUT_ADAPTIVE: ob.MUT_ADAPTIVE, nt.MUT_PERTURB_ONLY: ob.MUT_PERTURB_ONLY}[case["mutation_mode"]]
    kw = dict(seed=seed)
    if case.get("draws"):
        kw = dict(chain_u=case["draws"][0], swap_u=case["draws"][1])
    trace = case.get("trace", True)
    gpu = pu.gpu_run(wl, case["iters"], case["f"], mutation_mode=case["mutation_mode"],
                     weights=case["weights"], chunks=case["chunks"], trace=trace, **case["route"][1], **kw)
    orc = pu.oracle_run(wl, case["iters"], case["f"], mutation_mode=omut, weights=case["weights"], **kw)
    pu.assert_same_run(wl, gpu, orc, case["iters"], trace=trace)
    return gpu


# Without a step trace the incremental kernels may reject the removal of a search-tree edge from a lower
# bound of the cost (the node the edge led to is certainly lost) instead of counting everything that is
# lost with it.  Only runs that record no trace take that path, so it gets a sweep of its own: sparse
# graphs where most removals cut a tree edge, through the three incremental routes, compared with the
# oracle on the final state, the swap decisions, the slot map and the accepted / swap counts.
TRACEOFF_ROUTES = [
    ("incremental", dict(flags=0)),
    ("incremental-512", dict(flags=0, block_threads=512)),
    ("incremental-global", dict(flags=nt.FLAG_STREAMING_INCREMENTAL)),
    ("incremental-global-c2", dict(flags=nt.FLAG_STREAMING_INCREMENTAL, cluster_size=2)),
    ("incremental-global-c5", dict(flags=nt.FLAG_STREAMING_INCREMENTAL, cluster_size=5)),
]


def make_traceoff_case(rng, index):
    n = int(rng.choice([24, 65, int(rng.integers(30, 400)), int(rng.integers(400, 1500))]))
    T = int(rng.integers(2, 9))
    G = int(rng.integers(1, 3))
    iters = int(rng.integers(150, 500))
    f = int(rng.choice([1, 3, int(rng.integers(4, 60))]))
    deg = float(rng.choice([2.0, 2.02, 2.06, 2.1]))
    n_edges = max(int(n * deg / 2.0), n - 1)
    wl = wlm.graph_workload("fuzz-off", n, T, G, layout="box", avg_degree=deg,
                            edge_headroom=1.0 + (iters + 8) / max(n_edges, 1), index=index)
    # hot ladders propose topology moves four times out of five
    lo, hi = float(rng.choice([0.6, 2.5])), float(rng.choice([6.0, 12.0]))
    wl["temperatures"][:] = nt.geometric_ladder(lo, hi, T) if T > 1 else [hi]
    weights = None if rng.random() < 0.6 else [float(x) for x in rng.uniform(0.0, 2.5, size=6)]
    cut = int(rng.integers(0, iters + 1))
    route = TRACEOFF_ROUTES[int(rng.integers(0, len(TRACEOFF_ROUTES)))]
    return dict(wl=wl, iters=iters, f=f, mutation_mode=nt.MUT_ADAPTIVE, weights=weights,
                chunks=[c for c in (cut, iters - cut) if c > 0], route=route, draws=None, trace=False,
                summary=f"trace-off n={n} T={T} G={G} iters={iters} f={f} deg={deg} "
                        f"w={'custom' if weights else 'default'} route={route[0]}")


def make_chunk_case(rng, index):
    """The voxel-chunk engine's counterpart of make_case: terrain with a random share of air pockets,
    maybe an opened box around voxel 0 or an all-air / all-solid chunk, fractional densities,
    a random ladder, weights, swap interval and launch split."""
    chunks = int(rng.integers(1, 4))
    T = int(rng.integers(1, 7))
    iters = int(rng.integers(20, 250))
    f = int(rng.choice([1, 3, int(rng.integers(4, 60)), iters + 5]))
    wl = wlm.chunk_workload("fuzz", chunks, T, index=index, air_pockets=float(rng.choice([0.0, 0.03, 0.2, 0.6])))
    v = np.arange(32 ** 3)
    x, y, z = v % 32, (v // 32) % 32, v // 1024
    for g in range(chunks):
        m = rng.integers(0, 6)
        if m == 1:
            a, b, c = rng.integers(2, 20, size=3)
            wl["density"][g, :, (x < a) & (y < b) & (z < c)] = 0.0
        elif m == 2:
            wl["density"][g, :] = 0.0
        elif m == 3:
            wl["density"][g, :] = 1.0
        elif m == 4:                                # fractional densities around the solid threshold
            pick = rng.random(32 ** 3) < 0.3
            wl["density"][g, :, pick] = rng.random(int(pick.sum())).astype(np.float32)[:, None]
    tm = rng.integers(0, 3)
    if tm == 1:
        wl["temperatures"][:] = rng.uniform(0.05, 60.0, size=(chunks, T))
    elif tm == 2:
        wl["temperatures"][:] = np.sort(rng.uniform(5.0, 50.0, size=(chunks, T)), axis=1)
    weights = None if rng.random() < 0.5 else [float(w) for w in rng.uniform(0.0, 10.0, size=8)]
    cut = int(rng.integers(0, iters + 1))
    return dict(wl=wl, iters=iters, f=f, weights=weights, chunks=[c for c in (cut, iters - cut) if c > 0],
                summary=f"chunks={chunks} T={T} iters={iters} f={f} temps={tm} w={'custom' if weights else 'default'}")


def run_chunk_case(case, seed):
    import test_gpu_chunk_parity as cp
    gpu = cp._gpu(case["wl"], case["iters"], case["f"], seed=seed, weights=case["weights"], chunks=case["chunks"])
    cp._assert_same(case["wl"], gpu, case["iters"], case["f"], seed=seed, weights=case["weights"])


def make_collide_case(rng, index):
    """collision_detection: random clouds with clumps, coincident points, points at the bounds and far offsets."""
    n = int(rng.choice([0, 1, 2, 3, int(rng.integers(4, 300)), int(rng.integers(300, 6000))]))
    min_dist = float(rng.choice([0.5, 0.37, 0.05, 1.3, float(rng.uniform(0.002, 3.0))]))
    spread = min_dist * float(rng.choice([0.8, 2.0, 5.0, 20.0])) * max(n, 1) ** (1.0 / 3.0)
    pos = ((rng.random((n, 3)) - 0.5) * spread).astype(np.float32)
    if n >= 8:
        k = int(rng.integers(1, max(2, n // 4)))
        src, dst = rng.integers(0, n, size=k), rng.integers(0, n, size=k)
        jitter = rng.choice([0.0, 0.0005, 0.001, 0.5 * min_dist, min_dist, 1.0001 * min_dist], size=(k, 1))
        axis = np.eye(3, dtype=np.float64)[rng.integers(0, 3, size=k)]
        pos[dst] = (pos[src].astype(np.float64) + jitter * axis).astype(np.float32)
    if rng.random() < 0.3:
        pos += np.float32(rng.choice([-3.0e3, 1.0e5, 1.0e6]))
    return dict(pos=pos, min_dist=min_dist, summary=f"n={n} min_dist={min_dist} spread={spread:.3g}")


def run_collide_case(case, seed):
    import test_gpu_collision_detection as cd
    pos, md = case["pos"], case["min_dist"]
    want = ob.collision_pairs(pos, md)
    got, raw = cd._native(pos, md)
    assert np.array_equal(got, want), f"double-precision entry: {len(got)} pairs vs {len(want)}"
    assert (raw[2 * len(got):] == -7).all(), "wrote past the reported count"
    # the single-precision entry receives the bound rounded to fp32: that value is its bound
    md32 = float(np.float32(md))
    got32, _ = cd._native(pos, md32, f32=True)
    assert np.array_equal(got32, ob.collision_pairs(pos, md32)), "single-precision entry differs"


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    global make_case, run_case
    if len(sys.argv) > 3 and sys.argv[3] == "chunk":
        make_case, run_case = make_chunk_case, run_chunk_case
    if len(sys.argv) > 3 and sys.argv[3] == "large":
        global LARGE
        LARGE = True
    if len(sys.argv) > 3 and sys.argv[3] == "collide":
        make_case, run_case = make_collide_case, run_collide_case
    if len(sys.argv) > 3 and sys.argv[3] == "cluster":
        make_case = make_cluster_case
    early = [0]
    if len(sys.argv) > 3 and sys.argv[3] == "traceoff":
        graph_case = run_case

        def run_case(case, seed):                        # noqa: F811
            early[0] += graph_case(case, seed)["stats"]["early_rejects"]
        make_case = make_traceoff_case
    nt.load()
    failed = []
    t0 = time.time()
    for i in range(cases):
        rng = np.random.default_rng([seed0, i])
        case = make_case(rng, seed0 * 100000 + i)
        try:
            run_case(case, f"fuzz-{seed0}-{i}")
        except Exception as exc:                        # noqa: BLE001
            failed.append(i)
            print(f"FAIL case {i} (seed {seed0}): {case['summary']}\n  {type(exc).__name__}: {str(exc)[:600]}", flush=True)
            if not isinstance(exc, AssertionError):
                traceback.print_exc()
        if (i + 1) % (2 if LARGE else 20) == 0:
            print(f"{i + 1} cases, {len(failed)} failed, {time.time() - t0:.0f} s", flush=True)
    if len(sys.argv) > 3 and sys.argv[3] == "traceoff":
        print(f"removals rejected from the lower bound: {early[0]}", flush=True)
    print(f"done: {cases} cases, {len(failed)} failed {failed}", flush=True)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())

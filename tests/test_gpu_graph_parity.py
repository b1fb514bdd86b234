"""GPU parity tests of the spatial-graph engine: the HIP path (through the C ABI)
against the CPU oracle on identical inputs and identical random draws.

Bar: everything bit-exact against the oracle's canonical-sum mode (BFS counts,
proposals, accept decisions, swap decisions, positions, edge lists AND costs);
against the oracle's reference-order (sequential fp64) mode costs agree to
1e-12 relative.  The oracle itself is a restatement of the F# path pinned only
by the reference's scalar known-answer values (tests/test_oracle_kat.py); whole
trajectories are "parity unpinned" beyond that.
"""
import fractions

import numpy as np
import pytest

import oracle_binding as ob
import parity_util as pu
from procgen_amd import native as nt
from procgen_amd import workloads as wlm
from procgen_amd.engine import GraphEngine, canonical_sum

pytestmark = pytest.mark.gpu

# incremental evaluation (the default) and both memory placements of the full evaluation must
# give the same bits
PLACEMENTS = [pytest.param(0, id="incremental"), pytest.param(nt.FLAG_FULL_EVALUATION, id="full-lds"),
              pytest.param(nt.FLAG_GLOBAL_MEMORY_PLACEMENT, id="full-global"),
              pytest.param(nt.FLAG_STREAMING_INCREMENTAL, id="incremental-global")]
# ... and so must every number of workgroups cooperating on one replica's full evaluation
CLUSTERS = [pytest.param(c, id=f"cluster{c}") for c in (1, 2, 3, 4)]


def _exact_sum_py(x):
    tot = 0
    for t in x:
        fr = fractions.Fraction(float(t))
        q = int(abs(fr) * (1 << 64))          # trunc toward zero of |t| * 2^64
        tot += -q if fr < 0 else q
    return float(fractions.Fraction(tot, 1 << 64))


def test_square_root_of_the_node_passes_is_correctly_rounded_everywhere(gpu_lib):
    """The node passes take distances through a reciprocal-root approximation and two fused multiply-adds after
    vouching that the squared distance lies in [2^-20, 2^24): every float of that window (369 098 752 of them) is
    compared on the device with (float)sqrt((double)x), the correctly rounded root (sqrtf of the reference's
    Vector3.Distance, src/ParallelTempering.Core.fs:213, 236-240)."""
    import ctypes as C
    bad = C.c_uint64(12345)
    assert gpu_lib.pt_debug_sqrt_check(C.c_float(2.0 ** -20), C.c_float(2.0 ** 24), C.byref(bad)) == 0
    assert bad.value == 0
    # the hook does see a wrong root: outside the window (denormal results' neighbourhood) the sequence is not exact
    assert gpu_lib.pt_debug_sqrt_check(C.c_float(2.0 ** -126), C.c_float(2.0 ** -120), C.byref(bad)) == 0
    assert bad.value > 0


def test_fixed_point_sum_matches_big_integer_arithmetic(gpu_lib):
    rng = np.random.default_rng(1)
    cases = [
        rng.random(1000) * 1e-3,
        rng.standard_normal(5000) * 1e6,
        np.array([1e18, -1e18, 3.5, 2.0 ** -70, 2.0 ** -64, 2.0 ** -63, 5e-324, -5e-324]),
        np.array([0.1] * 777),
        rng.standard_normal(4096) * np.exp(rng.uniform(-60, 40, 4096)),
        np.array([]),
    ]
    for x in cases:
        got, err = canonical_sum(x)
        assert err == 0
        assert got == _exact_sum_py(x)
    _, err = canonical_sum(np.array([1.0, np.inf]))
    assert err == 1
    _, err = canonical_sum(np.array([2.0 ** 62]))
    assert err == 1


def _oracle_terms(wl, g, sum_mode, algo):
    cfg = ob.default_config(sum_mode, algo)
    ec = wl["ecount"][g, 0]
    return ob.cost(wl["pos"][g, 0], wl["edges"][g, 0, :ec], wl["kind"][g], wl["value"][g], cfg)


@pytest.mark.parametrize("flags", PLACEMENTS)
@pytest.mark.parametrize("n,layout,groups", [(1, "box", 1), (2, "box", 1), (20, "sphere", 1), (64, "box", 3),
                                             (300, "box", 2), (1000, "sphere", 1), (2500, "box", 1), (10000, "box", 1)])
def test_initial_cost_terms_bit_exact(gpu_lib, n, layout, groups, flags):
    wl = wlm.graph_workload("cost-terms", n, 3, groups, layout=layout)
    with GraphEngine(n, wl["ecap"], 3, groups, seed="t", flags=flags) as eng:
        eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
        terms = eng.cost_terms()
        cost = eng.download(positions=False, edges=False)["cost"]
    for g in range(groups):
        t = _oracle_terms(wl, g, ob.SUM_CANONICAL, ob.ALGO_OPTIMISED)
        want = [t.conn, t.endpoint, t.edge, t.coll, t.dist, t.trans, t.total, float(t.reach0)]
        for k in range(3):
            assert list(terms[g, k]) == want
            assert cost[g, k] == t.total
        # reference-order sums (sequential fp64, fp32 centroid fold) agree to rounding
        f = _oracle_terms(wl, g, ob.SUM_FSHARP, ob.ALGO_FAITHFUL if n <= 300 else ob.ALGO_OPTIMISED)
        assert f.reach0 == t.reach0 and f.npairs == t.npairs
        assert t.total == pytest.approx(f.total, rel=1e-9)
        assert t.edge == pytest.approx(f.edge, rel=1e-12, abs=1e-12)
        assert t.coll == pytest.approx(f.coll, rel=1e-12, abs=1e-12)
        assert t.trans == pytest.approx(f.trans, rel=1e-12, abs=1e-12)
        assert t.dist == pytest.approx(f.dist, rel=1e-5, abs=1e-9)   # fp32 centroid fold vs exact centroid


@pytest.mark.parametrize("flags", PLACEMENTS)
def test_empty_graph_no_edges_and_dense_collisions(gpu_lib, flags):
    # config 1's start: unit sphere, Edges = [] (reference ProceduralGeneration.fs:42); and a
    # collision-dense cloud where every node has many neighbours
    n = 400
    wl = wlm.graph_workload("dense", n, 2, 1, layout="sphere")
    wl["ecount"][:] = 0
    wl["pos"][:] *= 0.6
    with GraphEngine(n, wl["ecap"], 2, 1, seed="t", flags=flags) as eng:
        eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
        terms = eng.cost_terms()
    t = _oracle_terms(wl, 0, ob.SUM_CANONICAL, ob.ALGO_OPTIMISED)
    assert t.npairs > 1000
    assert list(terms[0, 0]) == [t.conn, t.endpoint, t.edge, t.coll, t.dist, t.trans, t.total, float(t.reach0)]
    assert t.conn == (n - 1) * 1000.0 and t.endpoint == 10000.0


@pytest.mark.parametrize("flags", PLACEMENTS)
def test_config1_trace_philox(gpu_lib, flags):
    """BASELINE config 1: 20 nodes, 5 temperatures 0.5-5.0, 1000 iterations, swap every 8,
    sphere start without edges (reference ProceduralGeneration.fs:95-128)."""
    n, T, iters, f = 20, 5, 1000, 8
    wl = wlm.graph_workload("config1", n, T, 1, layout="sphere")
    wl["ecount"][:] = 0
    gpu = pu.gpu_run(wl, iters, f, seed="config1", flags=flags)
    orc = pu.oracle_run(wl, iters, f, seed="config1")
    pu.assert_same_run(wl, gpu, orc, iters)
    tr = gpu["trace"]
    assert set(np.unique(tr["kind"])) == {0, 1, 2}
    assert (tr["flags"] & 4).any() and not (tr["flags"] & 4).all()
    assert gpu["swap_trace"].any() and not gpu["swap_trace"].all()
    assert gpu["stats"]["proposals"] == iters * T
    assert gpu["stats"]["accepted"] == int(((tr["flags"] & 4) != 0).sum())
    assert gpu["stats"]["swaps"] == int(gpu["swap_trace"].sum())
    # the reference-order oracle (sequential sums, reference algorithms) makes the same decisions here
    ref = pu.oracle_run(wl, iters, f, seed="config1", algo=ob.ALGO_FAITHFUL, sum_mode=ob.SUM_FSHARP)
    same = np.array_equal(ref[0]["trace"]["flags"], orc[0]["trace"]["flags"]) and \
        np.array_equal(ref[0]["swap_trace"], orc[0]["swap_trace"])
    assert same, "canonical-sum and reference-order oracles diverged on config 1"
    assert gpu["cost"][0] == pytest.approx(ref[0]["cost"], rel=1e-5)
    assert np.allclose(gpu["pos"][0], ref[0]["pos"], rtol=1e-5, atol=0)


@pytest.mark.parametrize("n,T,G,iters,f,layout", [(50, 4, 3, 400, 5, "box"), (300, 6, 2, 300, 34, "box"),
                                                   (1000, 16, 2, 130, 63, "box"), (200, 3, 1, 300, 7, "sphere")])
@pytest.mark.parametrize("flags", PLACEMENTS)
def test_trace_parity_philox(gpu_lib, n, T, G, iters, f, layout, flags):
    wl = wlm.graph_workload("trace", n, T, G, layout=layout, index=n)
    gpu = pu.gpu_run(wl, iters, f, seed=f"trace-{n}", chunks=[iters // 3, iters - iters // 3], flags=flags)
    orc = pu.oracle_run(wl, iters, f, seed=f"trace-{n}")
    pu.assert_same_run(wl, gpu, orc, iters)


@pytest.mark.parametrize("flags", PLACEMENTS)
@pytest.mark.parametrize("n,T,iters", [(60, 5, 400), (700, 6, 300)])
def test_trace_parity_seeded_generator_start(gpu_lib, n, T, iters, flags):
    """The starting graph of DeterministicGeneration.generate (reference src/Deterministic.fs:138-140,166):
    perturbed fibonacci sphere, hash-selected edges (nearly all of them leave the first few nodes),
    every replica starting from the same graph."""
    from procgen_amd import deterministic as det
    wl = det.deterministic_workload(f"seeded-{n}", n, T)
    f = nt.swap_frequency(n)
    gpu = pu.gpu_run(wl, iters, f, seed=f"seeded-{n}", flags=flags)
    orc = pu.oracle_run(wl, iters, f, seed=f"seeded-{n}")
    pu.assert_same_run(wl, gpu, orc, iters)


@pytest.mark.parametrize("cluster", CLUSTERS)
@pytest.mark.parametrize("n,T,G,iters,f", [(20, 5, 1, 300, 8), (300, 6, 2, 200, 34), (1000, 4, 2, 130, 63)])
def test_trace_parity_clustered(gpu_lib, n, T, G, iters, f, cluster):
    wl = wlm.graph_workload("cluster", n, T, G, index=n)
    gpu = pu.gpu_run(wl, iters, f, seed=f"cl-{n}", chunks=[iters // 2, iters - iters // 2], cluster_size=cluster,
                     flags=nt.FLAG_FULL_EVALUATION)
    orc = pu.oracle_run(wl, iters, f, seed=f"cl-{n}")
    pu.assert_same_run(wl, gpu, orc, iters)


def _topology_case(name):
    """Inputs that make the reached set change: what the incremental evaluation must track."""
    rng = np.random.default_rng(11)
    if name == "near-tree":         # path + 6 extra edges: most removals cut a tree edge
        n, T = 200, 4
        wl = wlm.graph_workload("topo-tree", n, T, 1, avg_degree=2.06, edge_headroom=3.0)
        temps = [8.0, 5.0, 3.0, 1.0]
    elif name == "fragments":       # no spanning path: few nodes reached, additions extend the reach
        n, T = 150, 4
        wl = wlm.graph_workload("topo-frag", n, T, 1, edge_headroom=3.0)
        e = rng.integers(0, n, size=(170, 2)).astype(np.int32)
        e = e[e[:, 0] != e[:, 1]]
        wl["edges"][:] = 0
        wl["edges"][0, :, :len(e)] = e
        wl["ecount"][:] = len(e)
        temps = [9.0, 6.0, 2.5, 0.7]
    elif name == "origin-elsewhere":    # the endpoint search starts at node 7, not node 0
        n, T = 120, 3
        wl = wlm.graph_workload("topo-origin", n, T, 1, avg_degree=2.2, edge_headroom=3.0)
        wl["kind"][0, 0] = nt.NODE_STANDARD
        wl["kind"][0, 7] = nt.NODE_ORIGIN
        wl["kind"][0, n - 1] = nt.NODE_STANDARD
        wl["kind"][0, 40] = nt.NODE_TERMINAL
        temps = [7.0, 4.0, 1.5]
    elif name == "duplicates":      # repeated and antiparallel edges in the uploaded list
        n, T = 60, 3
        wl = wlm.graph_workload("topo-dup", n, T, 1, avg_degree=2.4, edge_headroom=4.0)
        ec = int(wl["ecount"][0, 0])
        extra = np.concatenate([wl["edges"][0, 0, :6], wl["edges"][0, 0, 3:9, ::-1], wl["edges"][0, 0, :3]])
        wl["edges"][0, :, ec:ec + len(extra)] = extra
        wl["ecount"][:] = ec + len(extra)
        temps = [8.0, 4.0, 2.5]
    elif name == "dense":           # E > 3 N: the search's adjacency outgrows the position region
        n, T = 64, 3
        wl = wlm.graph_workload("topo-dense", n, T, 1, avg_degree=9.0, edge_headroom=2.0)
        temps = [6.0, 3.0, 1.0]
    elif name == "hub":             # 150 edges at node 5: more than the incremental kernel's list of incident edges holds
        n, T = 400, 3
        wl = wlm.graph_workload("topo-hub", n, T, 1, avg_degree=2.2, edge_headroom=3.0)
        ec = int(wl["ecount"][0, 0])
        others = rng.choice(np.arange(6, n), 150, replace=False).astype(np.int32)
        extra = np.stack([np.where(np.arange(150) % 2 == 0, 5, others), np.where(np.arange(150) % 2 == 0, others, 5)], axis=1)
        wl["edges"][0, :, ec:ec + 150] = extra
        wl["ecount"][:] = ec + 150
        # the hub is what moves most of the time: every replica perturbs, node 5 sits where many draws land
        temps = [1.5, 1.0, 0.6]
    elif name == "clump":           # every node within reach of every other: the list of near nodes overflows
        n, T = 1500, 3
        wl = wlm.graph_workload("topo-clump", n, T, 1, avg_degree=2.2, edge_headroom=2.0)
        wl["pos"][:] = (rng.random((n, 3)) * 0.9).astype(np.float32)
        temps = [1.2, 0.8, 0.5]
    else:
        raise KeyError(name)
    wl["temperatures"][:] = [temps]
    return wl


@pytest.mark.parametrize("flags", PLACEMENTS)
@pytest.mark.parametrize("case", ["near-tree", "fragments", "origin-elsewhere", "duplicates", "dense", "hub", "clump"])
def test_trace_parity_topology_cases(gpu_lib, case, flags):
    wl = _topology_case(case)
    iters, f = 600, 7
    gpu = pu.gpu_run(wl, iters, f, seed=case, chunks=[250, 350], flags=flags)
    orc = pu.oracle_run(wl, iters, f, seed=case)
    pu.assert_same_run(wl, gpu, orc, iters)
    reach = gpu["trace"]["reach0"]
    if case == "hub":
        assert ((gpu["trace"]["kind"] == 0) & (gpu["trace"]["a"] == 5)).any(), "the hub was meant to move"
    if case in ("near-tree", "fragments"):
        assert len(np.unique(reach)) > 3, "the case was meant to change the reached set"
    # the tracked cost is a from-scratch evaluation of the final state
    assert np.array_equal(gpu["terms"][0, :, 6], gpu["cost"][0])


@pytest.mark.parametrize("case", ["near-tree", "fragments", "hub", "duplicates"])
def test_incremental_kernel_without_incident_lists(gpu_lib, case, monkeypatch):
    """The incremental kernel keeps per-node lists of incident edges where they fit; without them (PROCGEN_PT_DELTA_LISTS=0)
    it prices a move from the edge registers and searches a cut-off subtree by pointer jumping: same trajectory."""
    monkeypatch.setenv("PROCGEN_PT_DELTA_LISTS", "0")
    wl = _topology_case(case)
    iters, f = 600, 7
    gpu = pu.gpu_run(wl, iters, f, seed=case, chunks=[250, 350])
    orc = pu.oracle_run(wl, iters, f, seed=case)
    pu.assert_same_run(wl, gpu, orc, iters)


@pytest.mark.parametrize("walk", ["0,128,1024", "3,128,1024", "256,2,1024", "256,128,2", "256,128,1024"],
                         ids=["never", "3-nodes", "2-entering-edges", "2-round-trips", "widest"])
@pytest.mark.parametrize("case", ["near-tree", "fragments"])
def test_subtree_walk_bounds_only_choose_the_search(gpu_lib, case, walk, monkeypatch):
    """A subtree cut off by a removal is walked through the lists within bounds (nodes, edges that enter it, round trips);
    beyond them the general search starts over.  Whatever the bounds, the trajectory is the oracle's."""
    monkeypatch.setenv("PROCGEN_PT_DELTA_WALK", walk)
    wl = _topology_case(case)
    iters, f = 600, 7
    gpu = pu.gpu_run(wl, iters, f, seed=case, chunks=[250, 350])
    orc = pu.oracle_run(wl, iters, f, seed=case)
    pu.assert_same_run(wl, gpu, orc, iters)
    assert len(np.unique(gpu["trace"]["reach0"])) > 3


@pytest.mark.parametrize("shape", ["mixed-300", "perturb-only-700", "mixed-1000-one-wave", "mixed-10000", "drifted-cloud"])
def test_untraced_run_ends_where_the_oracle_does(gpu_lib, shape):
    """Without a step trace the kernels take the paths they take in production (no per-step record, the early rejection
    of a removal that is bound to lose a node): final states, costs, accepted counts and swap decisions are the oracle's."""
    mode, block = nt.MUT_ADAPTIVE, 0
    if shape == "mixed-300":
        n, T, G, iters, f = 300, 6, 2, 500, 34
    elif shape == "perturb-only-700":
        n, T, G, iters, f, mode = 700, 5, 1, 600, 52, nt.MUT_PERTURB_ONLY
    elif shape == "mixed-1000-one-wave":
        n, T, G, iters, f, block = 1000, 4, 2, 400, 63, 64
    elif shape == "mixed-10000":
        n, T, G, iters, f = 10000, 4, 1, 450, 200
    else:
        n, T, G, iters, f = 400, 4, 1, 500, 40
    wl = wlm.graph_workload("untraced-" + shape, n, T, G, index=n)
    if shape == "drifted-cloud":        # far from the origin: coordinates and centroid lose low bits
        wl["pos"][:] += np.array([900.0, -350.0, 120.0], np.float32)
    gpu = pu.gpu_run(wl, iters, f, seed="untraced-" + shape, trace=False, mutation_mode=mode, block_threads=block,
                     chunks=[iters // 3, iters - iters // 3])
    orc = pu.oracle_run(wl, iters, f, seed="untraced-" + shape, mutation_mode=mode, threads=4)
    pu.assert_same_run(wl, gpu, orc, iters, trace=False)
    assert np.array_equal(gpu["terms"][:, :, 6], gpu["cost"]), "the tracked cost is not a from-scratch evaluation of the final state"


@pytest.mark.parametrize("block", [512, 768, 1024])
def test_trace_parity_incremental_kernel_variants(gpu_lib, block):
    """The incremental kernel holds 24, 32 or 48 edges per thread depending on the workgroup size."""
    n, T, iters, f = 10500, 4, 160, 7
    wl = wlm.graph_workload("variants", n, T, 1)
    assert 24 * 768 < wl["ecap"] <= 32 * 768 and 24 * 512 < wl["ecap"] <= 48 * 512
    gpu = pu.gpu_run(wl, iters, f, seed="variants", block_threads=block)
    orc = pu.oracle_run(wl, iters, f, seed="variants", threads=4)
    pu.assert_same_run(wl, gpu, orc, iters)


def test_trace_parity_beyond_16_bit_node_ids(gpu_lib):
    """70 000 nodes: no LDS placement, no packed 16-bit ids; the shape class of the million-node config."""
    n, T, iters, f = 70000, 3, 18, 5
    wl = wlm.graph_workload("wide-ids", n, T, 1)
    gpu = pu.gpu_run(wl, iters, f, seed="wide-ids")
    orc = pu.oracle_run(wl, iters, f, seed="wide-ids", threads=3)
    pu.assert_same_run(wl, gpu, orc, iters)


@pytest.mark.parametrize("flags", [0, nt.FLAG_STREAMING_INCREMENTAL], ids=["incremental", "incremental-global"])
def test_periodic_rebuild_of_the_search_state_changes_nothing(gpu_lib, flags, monkeypatch):
    """The incremental kernels' state is rebuilt from scratch now and then (their label counter must
    not wrap); with the rebuild forced every 37 steps the trajectory is still the oracle's."""
    monkeypatch.setenv("PROCGEN_PT_RELABEL_EVERY", "37")
    wl = _topology_case("near-tree")
    iters, f = 500, 7
    gpu = pu.gpu_run(wl, iters, f, seed="rebuild", chunks=[100, 150, 250], flags=flags)
    orc = pu.oracle_run(wl, iters, f, seed="rebuild")
    pu.assert_same_run(wl, gpu, orc, iters)
    assert len(np.unique(gpu["trace"]["reach0"])) > 3


@pytest.mark.parametrize("flags", [0, nt.FLAG_STREAMING_INCREMENTAL], ids=["incremental", "incremental-global"])
def test_rebuild_inside_a_swap_interval(gpu_lib, flags, monkeypatch):
    """A swap interval longer than the rebuild period (as at 10^6 nodes, where the interval is 2000 steps and the
    period about 800): the launch is cut into pieces with a rebuild between them."""
    monkeypatch.setenv("PROCGEN_PT_RELABEL_EVERY", "11")
    wl = _topology_case("near-tree")
    iters, f = 300, 50
    gpu = pu.gpu_run(wl, iters, f, seed="rebuild-inside", chunks=[120, 180], flags=flags)
    orc = pu.oracle_run(wl, iters, f, seed="rebuild-inside")
    pu.assert_same_run(wl, gpu, orc, iters)
    assert len(np.unique(gpu["trace"]["reach0"])) > 3


@pytest.mark.parametrize("route", [dict(flags=0), dict(flags=nt.FLAG_STREAMING_INCREMENTAL), dict(flags=nt.FLAG_STREAMING_INCREMENTAL, cluster_size=3)],
                         ids=["incremental", "incremental-global", "incremental-global-c3"])
def test_spent_label_budget_is_renewed_from_the_counters(gpu_lib, route, monkeypatch):
    """When the worst-case budget of discovery labels is spent the host reads how far the counters really got and goes
    on without a rebuild while most of the range is left (here the budget is 13 steps, so that happens dozens of times)."""
    monkeypatch.setenv("PROCGEN_PT_RELABEL_BUDGET", "13")
    wl = _topology_case("near-tree")
    iters, f = 400, 50
    gpu = pu.gpu_run(wl, iters, f, seed="renewed", chunks=[120, 280], **route)
    orc = pu.oracle_run(wl, iters, f, seed="renewed")
    pu.assert_same_run(wl, gpu, orc, iters)
    assert len(np.unique(gpu["trace"]["reach0"])) > 3


def test_trace_parity_many_small_replicas(gpu_lib):
    """More replicas than twice the CUs: the engine then picks its smallest workgroup."""
    n, T, G, iters, f = 300, 16, 40, 60, 7
    wl = wlm.graph_workload("many", n, T, G)
    gpu = pu.gpu_run(wl, iters, f, seed="many")
    orc = pu.oracle_run(wl, iters, f, seed="many", threads=4)
    pu.assert_same_run(wl, gpu, orc, iters)


@pytest.mark.parametrize("members", [2, 7, 16])
@pytest.mark.parametrize("case", ["near-tree", "fragments", "origin-elsewhere", "duplicates", "hub", "clump"])
def test_trace_parity_streaming_cluster(gpu_lib, case, members):
    """The global-memory incremental kernel with one replica's loops shared by several workgroups."""
    wl = _topology_case(case)
    iters, f = 400, 7
    gpu = pu.gpu_run(wl, iters, f, seed=case, chunks=[150, 250], flags=nt.FLAG_STREAMING_INCREMENTAL, cluster_size=members)
    orc = pu.oracle_run(wl, iters, f, seed=case)
    pu.assert_same_run(wl, gpu, orc, iters)
    assert np.array_equal(gpu["terms"][0, :, 6], gpu["cost"][0])


@pytest.mark.parametrize("members", [2, 16])
@pytest.mark.parametrize("self_loops", [False, True])
def test_streaming_cluster_incidence_lists_at_the_brim(gpu_lib, members, self_loops):
    """The clustered streaming kernel keeps per-node lists of incident edges with room for 16.  Here most nodes
    have about that many, on a hot ladder that mostly adds and removes edges, over several launches: lists fill up
    and overflow under accepted additions, overflowed nodes take the pass over the edge list, the next launch
    rebuilds them; uploaded self loops (which the lists leave out) can be drawn for removal."""
    n, T, iters, f = 40, 4, 900, 9
    wl = wlm.graph_workload("brim", n, T, 1, avg_degree=15.0, edge_headroom=2.0)
    if self_loops:
        ec = int(wl["ecount"][0, 0])
        extra = np.array([[3, 3], [17, 17], [3, 3], [0, 0], [39, 39]], np.int32)
        wl["edges"][0, :, ec:ec + len(extra)] = extra
        wl["ecount"][:] = ec + len(extra)
    wl["temperatures"][:] = [[9.0, 6.0, 3.5, 2.2]]
    deg = np.bincount(wl["edges"][0, 0, :wl["ecount"][0, 0]].ravel(), minlength=n)
    assert (deg <= 16).sum() >= 5 and (deg > 16).sum() >= 5
    gpu = pu.gpu_run(wl, iters, f, seed="brim", chunks=[100, 350, 450], flags=nt.FLAG_STREAMING_INCREMENTAL, cluster_size=members)
    orc = pu.oracle_run(wl, iters, f, seed="brim")
    pu.assert_same_run(wl, gpu, orc, iters)
    tr = gpu["trace"]
    assert ((tr["kind"] == 1) & ((tr["flags"] & 5) == 4)).sum() > 50, "additions were meant to be accepted"
    assert ((tr["kind"] == 2) & ((tr["flags"] & 5) == 4)).sum() > 50, "removals were meant to be accepted"
    assert np.array_equal(gpu["terms"][0, :, 6], gpu["cost"][0])


@pytest.mark.parametrize("runs", [0, 37])
def test_streaming_cluster_distance_cache_split(gpu_lib, runs, monkeypatch):
    """The clustered streaming kernel keeps the distances of the first distribution pass in LDS as far as they fit and
    sends the rest through global memory; at test sizes everything fits, so the room is cut down here (to nothing,
    and to 37 runs of four, which splits the first wave's nodes between the two routes)."""
    monkeypatch.setenv("PROCGEN_PT_BIGC_CACHE_RUNS", str(runs))
    wl = _topology_case("clump")
    iters, f = 300, 7
    gpu = pu.gpu_run(wl, iters, f, seed="clump", flags=nt.FLAG_STREAMING_INCREMENTAL, cluster_size=2)
    orc = pu.oracle_run(wl, iters, f, seed="clump")
    pu.assert_same_run(wl, gpu, orc, iters)


def test_streaming_cluster_matches_one_workgroup_at_150k_nodes(gpu_lib):
    """Large enough for every member to stream thousands of nodes: the trace of the clustered kernel
    equals the single-workgroup kernel's (which the oracle pins at 70 000 nodes)."""
    n, T, iters, f = 150000, 6, 240, 20
    wl = wlm.graph_workload("wide-cluster", n, T, 1)
    one = pu.gpu_run(wl, iters, f, seed="wide-cluster", cluster_size=1)
    many = pu.gpu_run(wl, iters, f, seed="wide-cluster", cluster_size=16)
    for k in pu.TRACE_FIELDS:
        assert np.array_equal(one["trace"][k], many["trace"][k]), k
    assert np.array_equal(one["swap_trace"], many["swap_trace"])
    for k in ("cost", "pos", "edges", "ecount"):
        assert np.array_equal(one[k], many[k]), k
    assert one["stats"]["status_bits"] == 0 and many["stats"]["status_bits"] == 0
    assert (one["trace"]["kind"] == 2).any() and len(np.unique(one["trace"]["reach0"])) > 1


def test_million_node_replicas_track_their_cost(gpu_lib):
    """BASELINE configs[4]'s replica size: 10^6 nodes (two replicas here), 16 workgroups per replica.  After a
    few dozen proposals the tracked cost of each replica equals a from-scratch evaluation on the device and
    the oracle's evaluation of the downloaded state, bit for bit, and the run equals the one-workgroup
    kernel's (whose trajectories the oracle pins at 70 000 nodes)."""
    n, T, iters, f = 1000000, 2, 36, 16
    wl = wlm.graph_workload("c5-test", n, T, 1)
    wl["temperatures"][:] = [[5.0, 0.5]]
    runs = []
    for members in (0, 1):
        with GraphEngine(n, wl["ecap"], T, 1, swap_frequency=f, seed="c5-test", cluster_size=members) as eng:
            info = eng.info()
            assert info["incremental"] and not info["lds_placement"]
            assert (info["cluster_size"] == 16) if members == 0 else (info["cluster_size"] == 1)
            eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
            eng.run(iters)
            res = eng.download()
            res["terms"], res["stats"], res["slots"] = eng.cost_terms(), eng.stats(), eng.replica_slots()
            runs.append(res)
    many, one = runs
    assert many["stats"]["status_bits"] == 0 and one["stats"]["status_bits"] == 0
    assert many["stats"]["proposals"] == iters * T and 0 < many["stats"]["accepted"] < iters * T
    for k in ("cost", "ecount", "pos", "edges", "slots"):
        assert np.array_equal(many[k], one[k]), k
    assert many["stats"]["accepted"] == one["stats"]["accepted"] and many["stats"]["swaps"] == one["stats"]["swaps"]
    assert np.array_equal(many["terms"][0, :, 6], many["cost"][0])
    cfg = ob.default_config(ob.SUM_CANONICAL, ob.ALGO_OPTIMISED)
    for t in range(T):
        ec = many["ecount"][0, t]
        o = ob.cost(many["pos"][0, t], many["edges"][0, t, :ec], wl["kind"][0], wl["value"][0], cfg)
        assert o.total == many["cost"][0, t] and float(o.reach0) == many["terms"][0, t, 7]


def test_trace_parity_injected_draws(gpu_lib):
    n, T, G, iters, f = 120, 5, 2, 500, 6
    wl = wlm.graph_workload("inject", n, T, G)
    rng = np.random.default_rng(7)
    passes = (iters + f - 1) // f
    chain_u = rng.random((G, T, iters, nt.DRAWS_PER_STEP))
    swap_u = rng.random((G, passes, T - 1))
    # force the rare branches: exact 0.0 draws and draws just below 1
    chain_u[0, 0, 3, :] = 0.0
    chain_u[1, 2, 5, :] = np.nextafter(1.0, 0.0)
    gpu = pu.gpu_run(wl, iters, f, chain_u=chain_u, swap_u=swap_u)
    orc = pu.oracle_run(wl, iters, f, chain_u=chain_u, swap_u=swap_u)
    pu.assert_same_run(wl, gpu, orc, iters)


def test_ladder_keys_make_a_ladder_independent_of_its_batch_index(gpu_lib):
    """With per-ladder keys a ladder's draws depend on its key, not on where it sits in the batch: two
    batches holding the same ladders in different order give the same traces, which are the oracle's."""
    n, T, G, iters, f = 90, 4, 3, 200, 7
    wl = wlm.graph_workload("keyed", n, T, G)
    base = nt.seed_key("keyed-seed")
    keys = np.array([nt.hash_coordinates(base, 5 * g - 3, g, -g) for g in range(G)], np.uint64)
    order_b = [1, 2, 0]
    runs = []
    for order in (list(range(G)), order_b):
        with GraphEngine(n, wl["ecap"], T, G, swap_frequency=f, seed="keyed-seed") as eng:
            eng.upload(wl["pos"][order], wl["edges"][order], wl["ecount"][order], wl["temperatures"][order],
                       wl["kind"][order], wl["value"][order])
            eng.set_ladder_keys(keys[order])
            passes = (iters + f - 1) // f
            eng.enable_trace(iters, passes)
            eng.run(iters)
            res = eng.download()
            res["trace"], res["swap_trace"] = eng.read_trace(iters, passes)
            assert eng.stats()["status_bits"] == 0
            runs.append(res)
    a, b = runs
    cfg = ob.default_config(ob.SUM_CANONICAL, ob.ALGO_OPTIMISED)
    for g in range(G):
        gb = order_b.index(g)
        o = ob.run(wl["pos"][g], wl["edges"][g], wl["ecount"][g], wl["kind"][g], wl["value"][g], wl["temperatures"][g],
                   iters, f, cfg, key=int(keys[g]), stream_base=0, swap_stream_id=0)
        for fld in pu.TRACE_FIELDS:
            assert np.array_equal(a["trace"][:, g * T:(g + 1) * T][fld], b["trace"][:, gb * T:(gb + 1) * T][fld]), fld
            assert np.array_equal(a["trace"][:, g * T:(g + 1) * T][fld], o["trace"][fld]), fld
        assert np.array_equal(a["swap_trace"][:, g], o["swap_trace"]) and np.array_equal(b["swap_trace"][:, gb], o["swap_trace"])
        assert np.array_equal(a["cost"][g], o["cost"]) and np.array_equal(a["pos"][g], o["pos"])


def test_perturb_only_and_custom_weights(gpu_lib):
    n, T, iters, f = 150, 4, 300, 9
    wl = wlm.graph_workload("perturb", n, T, 1)
    w = [0.5, 2.0, 0.3, 1.5, 0.2, 0.7]
    gpu = pu.gpu_run(wl, iters, f, seed="p", mutation_mode=nt.MUT_PERTURB_ONLY, weights=w)
    orc = pu.oracle_run(wl, iters, f, seed="p", mutation_mode=ob.MUT_PERTURB_ONLY, weights=w)
    pu.assert_same_run(wl, gpu, orc, iters)
    assert (gpu["trace"]["kind"] == 0).all()


def test_edge_capacity_hit_is_reported(gpu_lib):
    n, T, iters, f = 30, 3, 300, 5
    wl = wlm.graph_workload("cap", n, T, 1)
    wl["temperatures"][:] = [[9.0, 7.0, 5.0]]           # hot: mostly topology moves
    ecap = int(wl["ecount"].max()) + 2
    wl["ecap"] = ecap
    wl["edges"] = np.ascontiguousarray(wl["edges"][:, :, :ecap])
    eng = GraphEngine(n, ecap, T, 1, swap_frequency=f, seed="cap")
    eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
    eng.enable_trace(iters, iters)
    eng.run(iters)
    tr, _ = eng.read_trace(iters, 0)
    st = eng.stats()
    res = eng.download()
    eng.close()
    orc = pu.oracle_run(wl, iters, f, seed="cap")
    assert np.array_equal(tr["flags"], orc[0]["trace"]["flags"])
    assert (tr["flags"] & 8).any()
    assert st["status_bits"] == 1 << 5
    assert np.array_equal(res["cost"][0], orc[0]["cost"])


@pytest.mark.parametrize("cluster", [2, 4])
def test_10k_nodes_clustered_matches_oracle(gpu_lib, cluster):
    n, T, iters, f = 10000, 4, 24, 200
    wl = wlm.graph_workload("c2-small", n, T, 1)
    wl["temperatures"][:] = [[5.0, 2.5, 1.0, 0.5]]
    gpu = pu.gpu_run(wl, iters, f, seed="c2", cluster_size=cluster, flags=nt.FLAG_FULL_EVALUATION)
    orc = pu.oracle_run(wl, iters, f, seed="c2")
    pu.assert_same_run(wl, gpu, orc, iters)


@pytest.mark.parametrize("flags", PLACEMENTS)
def test_10k_nodes_short_run_matches_oracle(gpu_lib, flags):
    """The headline shape (10k nodes, E = 15k) on 4 replicas for a few dozen steps."""
    n, T, iters, f = 10000, 4, 24, 200
    wl = wlm.graph_workload("c2-small", n, T, 1)
    wl["temperatures"][:] = [[5.0, 2.5, 1.0, 0.5]]
    gpu = pu.gpu_run(wl, iters, f, seed="c2", flags=flags)
    orc = pu.oracle_run(wl, iters, f, seed="c2")
    pu.assert_same_run(wl, gpu, orc, iters)


def test_full_size_trajectory_matches_oracle(gpu_lib):
    """10k nodes x 64 replicas (BASELINE config 2), the benchmark's own route (no step trace) for 402
    iterations = three swap passes: the final positions, edge lists, costs, swap decisions, slot map
    and the accepted count of all 64 replicas equal the oracle's, bit for bit."""
    n, T, iters, f = 10000, 64, 402, 200
    wl = wlm.graph_workload("c2", n, T, 1)
    gpu = pu.gpu_run(wl, iters, f, seed="c2-full", trace=False, chunks=[1, 200, 201])
    orc = pu.oracle_run(wl, iters, f, seed="c2-full", threads=16)
    pu.assert_same_run(wl, gpu, orc, iters, trace=False)
    assert gpu["swap_trace"].shape[0] == 3 and gpu["swap_trace"].any()
    assert np.array_equal(gpu["terms"][0, :, 6], gpu["cost"][0])


def test_full_size_properties(gpu_lib):
    """10k nodes x 64 replicas (BASELINE config 2): size-independent checks.  After a run the
    tracked energy of every replica equals a from-scratch re-evaluation of its final state,
    bit for bit; the slot map is a permutation; downloading and re-uploading reproduces the
    same energies (idempotence)."""
    n, T, iters = 10000, 64, 402
    wl = wlm.graph_workload("c2", n, T, 1)
    eng = GraphEngine(n, wl["ecap"], T, 1, seed="c2-full")
    assert eng.info()["incremental"]
    eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
    eng.run(iters)
    res = eng.download()
    terms = eng.cost_terms()
    st = eng.stats()
    slots = eng.replica_slots()
    assert st["status_bits"] == 0
    assert st["proposals"] == iters * T and 0 < st["accepted"] < st["proposals"]
    assert st["swap_attempts"] == 3 * (T - 1)
    assert sorted(slots[0].tolist()) == list(range(T))
    assert np.array_equal(terms[0, :, 6], res["cost"][0])
    # spot-check three replicas against the oracle's cost of the downloaded state
    cfg = ob.default_config(ob.SUM_CANONICAL, ob.ALGO_OPTIMISED)
    for t in (0, 31, 63):
        ec = res["ecount"][0, t]
        o = ob.cost(res["pos"][0, t], res["edges"][0, t, :ec], wl["kind"][0], wl["value"][0], cfg)
        assert o.total == res["cost"][0, t] and float(o.reach0) == terms[0, t, 7]
    eng.upload(res["pos"], res["edges"], res["ecount"], wl["temperatures"], wl["kind"], wl["value"])
    again = eng.download(positions=False, edges=False)["cost"]
    eng.close()
    assert np.array_equal(again, res["cost"])

"""GPU parity tests of the voxel-chunk engine: the HIP path (through the C ABI) against the
terrain oracle on identical inputs and identical draws.  Bar: bit-exact against the oracle's
canonical-sum mode (every trace field, swap decision, voxel field and cost).  The oracle is this
build's restatement of the F# terrain path, checked only against hand-derived values
(tests/test_terrain_oracle.py): parity unpinned beyond that."""
import numpy as np
import pytest

import oracle_binding as ob
import terrain_binding as tb
from procgen_amd import native as nt
from procgen_amd import workloads as wlm
from procgen_amd.engine import ChunkEngine

pytestmark = pytest.mark.gpu
FIELDS = ["kind", "a", "b", "flags", "reach0", "reach_origin", "delta", "u", "cost_after"]

# Every way the step kernel is launched (pt_chunk_launch_shape): 64-lane workgroups holding 8, 16, 32 or 64
# replicas (8, 4, 2, 1 lanes per replica: draws shared four ways, shared two ways, not shared; 7, 3, 1, 0 helper lanes
# in the wave-wide cave search) and 256-lane workgroups.  The host picks by replica count -- 4096 chunks x 6
# temperatures run 32 per workgroup, 65 536 replicas and more the 256-lane form -- so the small cases below force each.
ROUTES = {"rpb8": ({"PROCGEN_PT_CHUNK_RPB": "8"}, (64, 8)), "rpb16": ({"PROCGEN_PT_CHUNK_RPB": "16"}, (64, 16)),
          "rpb32": ({"PROCGEN_PT_CHUNK_RPB": "32"}, (64, 32)), "rpb64": ({"PROCGEN_PT_CHUNK_RPB": "64"}, (64, 64)),
          "wide256": ({"PROCGEN_PT_CHUNK_WIDE": "1"}, (256, 256))}


@pytest.fixture(params=list(ROUTES))
def route(request, monkeypatch):
    """Sets the launch route for engines created inside the test; yields the (lanes, replicas) per workgroup expected."""
    for k in ("PROCGEN_PT_CHUNK_RPB", "PROCGEN_PT_CHUNK_WIDE", "PROCGEN_PT_CHUNK_WPS"):
        monkeypatch.delenv(k, raising=False)
    env, shape = ROUTES[request.param]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    return shape


def _oracle(wl, g, iters, f, *, seed=None, chain_u=None, swap_u=None, weights=None, T=None):
    T = wl["temps"]
    cfg = tb.default_config(ob.SUM_CANONICAL, weights=weights)
    kw = {}
    if chain_u is not None:
        kw = dict(chain_u=chain_u[g].reshape(T, -1), swap_u=swap_u[g].reshape(-1))
    else:
        kw = dict(key=int(ob.lib().og_seed_key(seed.encode())), stream_base=g * T, swap_stream_id=g)
    return tb.run(wl["density"][g], wl["moisture"][g], wl["temperature"][g], wl["material"][g], wl["temperatures"][g],
                  iters, f, cfg, **kw)


def _gpu(wl, iters, f, *, seed=None, chain_u=None, swap_u=None, weights=None, chunks=None, shape=None):
    eng = ChunkEngine(wl["temps"], wl["chunks"], swap_frequency=f, seed=seed, weights=weights,
                      rng_mode=nt.RNG_INJECTED if chain_u is not None else nt.RNG_PHILOX)
    try:
        if shape is not None:
            assert eng.launch_shape() == shape, "the launch route the test meant to force is not the one that runs"
        eng.upload(wl["density"], wl["moisture"], wl["temperature"], wl["material"], wl["temperatures"])
        init = eng.download(fields=False)["cost"]
        init_terms = eng.cost_terms()
        if chain_u is not None:
            eng.set_draws(chain_u, swap_u)
        passes = (iters + f - 1) // f
        eng.enable_trace(iters, passes)
        for c in (chunks or [iters]):
            eng.run(c)
        eng.sync()
        res = eng.download()
        res["trace"], res["swap_trace"] = eng.read_trace(iters, passes)
        res["terms"] = eng.cost_terms()
        res["stats"] = eng.stats()
        res["init_cost"], res["init_terms"] = init, init_terms
        return res
    finally:
        eng.close()


def _assert_same(wl, gpu, iters, f, **okw):
    T = wl["temps"]
    for g in range(wl["chunks"]):
        o = _oracle(wl, g, iters, f, **okw)
        assert o["err"] == 0
        gt = gpu["trace"][:, g * T:(g + 1) * T]
        first = None
        for fld in FIELDS:
            if not np.array_equal(gt[fld], o["trace"][fld]):
                bad = np.argwhere(gt[fld] != o["trace"][fld])[0]
                if first is None or (bad[0], bad[1]) < first[:2]:
                    first = (bad[0], bad[1], fld)
        if first is not None:
            s0, c0, fld = first
            raise AssertionError(f"chunk {g}: trace first differs at step {s0} chain {c0} (field {fld!r}):\n"
                                 f"  fields {FIELDS}\n  gpu    {gt[s0, c0]}\n  oracle {o['trace'][s0, c0]}")
        assert np.array_equal(gpu["swap_trace"][:, g, :], o["swap_trace"])
        assert np.array_equal(gpu["cost"][g], o["cost"])
        for name in ("density", "moisture", "temperature"):
            assert np.array_equal(gpu[name][g], o[name]), f"chunk {g}: {name} differs"
    assert gpu["stats"]["status_bits"] == 0
    # the tracked costs are from-scratch evaluations of the final states
    assert np.array_equal(gpu["terms"][:, :, 8], gpu["cost"])


def test_initial_terms_match_the_oracle(gpu_lib):
    wl = wlm.chunk_workload("init", 3, 2)
    wl["density"][1, :, 0] = 0.0                      # chunk 1: voxel 0 is air, the cave search has something to do
    wl["density"][1, :, 1] = 0.0
    wl["density"][1, :, 32] = 0.0
    wl["density"][2, :] = 0.0                         # chunk 2: all air
    with ChunkEngine(2, 3, seed="t") as eng:
        eng.upload(wl["density"], wl["moisture"], wl["temperature"], wl["material"], wl["temperatures"])
        terms = eng.cost_terms()
        cost = eng.download(fields=False)["cost"]
    cfg = tb.default_config(ob.SUM_CANONICAL)
    for g in range(3):
        t = tb.cost(wl["density"][g, 0], wl["moisture"][g, 0], wl["material"][g], cfg)
        want = [t.trav, t.geo, t.overhang, t.erosion, t.aesthetics, t.caves, t.performance, t.biome, t.total,
                float(t.reached), float(t.underground_air), float(t.solid)]
        for k in range(2):
            assert list(terms[g, k]) == want
            assert cost[g, k] == t.total
    assert terms[2, 0, 9] == 32 ** 3 and terms[1, 0, 9] >= 3


@pytest.mark.parametrize("chunks,T,iters,f", [(2, 6, 400, 50), (3, 3, 300, 7)])
def test_trace_parity_philox(gpu_lib, route, chunks, T, iters, f):
    wl = wlm.chunk_workload("trace", chunks, T, index=T)
    gpu = _gpu(wl, iters, f, seed=f"chunk-{T}", chunks=[iters // 3, iters - iters // 3], shape=route)
    _assert_same(wl, gpu, iters, f, seed=f"chunk-{T}")
    tr = gpu["trace"]
    assert set(np.unique(tr["kind"])) == {0, 1}
    assert (tr["flags"] & 4).any() and not (tr["flags"] & 4).all()
    assert gpu["stats"]["proposals"] == iters * chunks * T


def test_default_route_of_a_small_batch(gpu_lib, monkeypatch):
    for k in ("PROCGEN_PT_CHUNK_RPB", "PROCGEN_PT_CHUNK_WIDE", "PROCGEN_PT_CHUNK_WPS"):
        monkeypatch.delenv(k, raising=False)
    with ChunkEngine(6, 2, seed="t") as eng:
        assert eng.launch_shape() == (64, 8)


def test_more_replicas_than_one_workgroup_holds(gpu_lib, route):
    """Seven chunks x 6 temperatures = 42 replicas: the last workgroup of every 64-lane route is ragged (lanes past the
    last replica, helpers of lanes that hold none), the 64-per-workgroup and 256-lane routes have idle lanes."""
    wl = wlm.chunk_workload("ragged", 7, 6, index=3)
    gpu = _gpu(wl, 120, 25, seed="ragged", shape=route)
    _assert_same(wl, gpu, 120, 25, seed="ragged")


def test_cave_search_is_exercised(gpu_lib, route):
    """An open chunk (air around voxel 0, hot ladder): flips keep cutting and re-opening reached paths."""
    wl = wlm.chunk_workload("caves", 2, 4)
    v = np.arange(32 ** 3)
    x, y, z = v % 32, (v // 32) % 32, v // 1024
    open_region = (x < 12) & (y < 12) & (z < 12)
    wl["density"][:, :, open_region] = 0.0
    wl["temperatures"][:] = [[40.0, 30.0, 20.0, 10.0]]
    # make the search matter less than everything else would forbid: huge costs at T = 40 still reject most cuts
    weights = [8.0, 3.0, 12.0, 2.0, 6.0, 0.004, 10.0, 2.0]
    rng = np.random.default_rng(5)
    iters, f = 500, 25
    passes = (iters + f - 1) // f
    chain_u = rng.random((2, 4, iters, nt.CHUNK_DRAWS_PER_STEP))
    swap_u = rng.random((2, passes, 3))
    # aim the proposals at the open region and make most of them flips
    tx, ty, tz = (rng.integers(0, 12, size=(2, 4, iters)) for _ in range(3))
    chain_u[..., 0] = ((tx + 32 * ty + 1024 * tz) + 0.5) / 32 ** 3
    chain_u[..., 1] = np.where(rng.random((2, 4, iters)) < 0.8, 0.9, 0.1)
    chain_u[0, 0, 3, 0] = 0.0                          # voxel 0 itself
    chain_u[0, 0, 3, 1] = 0.9
    gpu = _gpu(wl, iters, f, chain_u=chain_u, swap_u=swap_u, weights=weights, shape=route)
    _assert_same(wl, gpu, iters, f, chain_u=chain_u, swap_u=swap_u, weights=weights)
    assert len(np.unique(gpu["trace"]["reach0"])) > 10, "the case was meant to change the cave reach"


def test_cave_search_philox_hot_open_chunks(gpu_lib, route):
    """The same kind of chunk under the engine's own draws (the shared Philox blocks of the 2- and 4-lane routes), several
    replicas of a wave needing the wave-wide search in the same step."""
    wl = wlm.chunk_workload("caves-philox", 4, 6)
    v = np.arange(32 ** 3)
    x, y, z = v % 32, (v // 32) % 32, v // 1024
    wl["density"][:, :, (x < 10) & (y < 10) & (z < 10)] = 0.0
    wl["density"][2:, :, (y < 16)] = 0.0                    # two chunks open all the way across: long cones
    wl["temperatures"][:] = [[60.0, 40.0, 25.0, 12.0, 6.0, 3.0]]
    weights = [8.0, 3.0, 12.0, 2.0, 6.0, 0.002, 10.0, 2.0]
    iters, f = 600, 50
    gpu = _gpu(wl, iters, f, seed="caves-philox", weights=weights, shape=route)
    _assert_same(wl, gpu, iters, f, seed="caves-philox", weights=weights)
    assert len(np.unique(gpu["trace"]["reach0"])) > 10, "the case was meant to change the cave reach"


def test_custom_weights_and_mutation_constants(gpu_lib):
    wl = wlm.chunk_workload("weights", 1, 4)
    w = [1.0, 5.0, 2.0, 7.0, 3.0, 0.5, 4.0, 6.0]
    gpu = _gpu(wl, 250, 11, seed="w", weights=w)
    _assert_same(wl, gpu, 250, 11, seed="w", weights=w)


def test_a_chunk_keyed_by_its_position_does_not_depend_on_its_place_in_the_batch(gpu_lib):
    """Terrain.Generation.generate seeds a chunk with hashCoordinates seed x y z (reference
    src/Terrain.Generation.fs:172-176).  With positions given, a chunk's draws depend on (seed, position) only:
    the same chunk at two places of two different batches has the same trace, and that trace is the oracle's
    for the key og_hash_coordinates gives."""
    T, iters, f, seed = 4, 150, 9, "terrain-seed"
    wl = wlm.chunk_workload("keyed", 3, T)
    pos_a = np.array([[4, -2, 7], [0, 0, 0], [-11, 3, 5]], np.int32)
    order_b = [2, 0, 1]                                  # the same chunks, shuffled
    runs = []
    for order, positions in ((list(range(3)), pos_a), (order_b, pos_a[order_b])):
        eng = ChunkEngine(T, 3, swap_frequency=f, seed=seed)
        try:
            eng.upload(wl["density"][order], wl["moisture"][order], wl["temperature"][order], wl["material"][order],
                       wl["temperatures"][order])
            eng.set_positions(positions)
            passes = (iters + f - 1) // f
            eng.enable_trace(iters, passes)
            eng.run(iters)
            eng.sync()
            res = eng.download()
            res["trace"], res["swap_trace"] = eng.read_trace(iters, passes)
            assert eng.stats()["status_bits"] == 0
            runs.append(res)
        finally:
            eng.close()
    a, b = runs
    base = int(ob.lib().og_seed_key(seed.encode()))
    cfg = tb.default_config(ob.SUM_CANONICAL)
    for g in range(3):
        gb = order_b.index(g)                            # where chunk g sits in the second batch
        for fld in FIELDS:
            assert np.array_equal(a["trace"][:, g * T:(g + 1) * T][fld], b["trace"][:, gb * T:(gb + 1) * T][fld]), fld
        assert np.array_equal(a["swap_trace"][:, g], b["swap_trace"][:, gb])
        assert np.array_equal(a["density"][g], b["density"][gb]) and np.array_equal(a["cost"][g], b["cost"][gb])
        key = int(ob.lib().og_hash_coordinates(base, int(pos_a[g, 0]), int(pos_a[g, 1]), int(pos_a[g, 2])))
        assert key == nt.hash_coordinates(base, *[int(v) for v in pos_a[g]])
        o = tb.run(wl["density"][g], wl["moisture"][g], wl["temperature"][g], wl["material"][g], wl["temperatures"][g],
                   iters, f, cfg, key=key, stream_base=0, swap_stream_id=0)
        for fld in FIELDS:
            assert np.array_equal(a["trace"][:, g * T:(g + 1) * T][fld], o["trace"][fld]), fld
        assert np.array_equal(a["swap_trace"][:, g], o["swap_trace"]) and np.array_equal(a["cost"][g], o["cost"])
    # two chunks with equal contents at different positions take different trajectories
    assert not np.array_equal(a["trace"][:, 0:T]["a"], a["trace"][:, T:2 * T]["a"])


def test_keys_cannot_change_in_mid_run(gpu_lib):
    wl = wlm.chunk_workload("keys", 1, 2)
    with ChunkEngine(2, 1, seed="k") as eng:
        eng.upload(wl["density"], wl["moisture"], wl["temperature"], wl["material"], wl["temperatures"])
        eng.set_ladder_keys([5])
        eng.run(3)
        with pytest.raises(nt.PtError):
            eng.set_ladder_keys([6])
        with pytest.raises(nt.PtError):
            eng.set_positions([[1, 2, 3]])
        eng.upload(wl["density"], wl["moisture"], wl["temperature"], wl["material"], wl["temperatures"])
        eng.set_positions([[1, 2, 3]])                   # a new run: allowed again


@pytest.mark.timeout(900)
def test_batched_shape_4096_chunks_tracked_costs_and_sampled_oracle_parity(gpu_lib, monkeypatch):
    """The batched terrain shape (4096 chunks x 6 temperatures = 24 576 replicas) through the route the host picks for
    it: after three swap intervals the tracked cost of every replica equals a from-scratch evaluation of its final
    state, and for a sample of the chunks the whole trace, the swap decisions and the final costs equal the oracle's."""
    for k in ("PROCGEN_PT_CHUNK_RPB", "PROCGEN_PT_CHUNK_WIDE", "PROCGEN_PT_CHUNK_WPS"):
        monkeypatch.delenv(k, raising=False)
    G, T, distinct, f = 4096, 6, 32, 50
    iters = 3 * f + 1
    passes = (iters + f - 1) // f
    wl = wlm.chunk_workload("batched", distinct, T, index=9)
    tile = lambda a: np.tile(a, (G // distinct,) + (1,) * (a.ndim - 1))     # noqa: E731
    seed = "batched-4096"
    eng = ChunkEngine(T, G, swap_frequency=f, seed=seed)
    try:
        assert eng.launch_shape() == (64, 32), "4096 x 6 is expected to run 32 replicas per 64-lane workgroup"
        eng.upload(tile(wl["density"]), tile(wl["moisture"]), tile(wl["temperature"]), tile(wl["material"]),
                   tile(wl["temperatures"]))
        eng.enable_trace(iters, passes)
        eng.run(iters)
        eng.sync()
        cost = eng.download(fields=False)["cost"]
        terms = eng.cost_terms()
        trace, swaps = eng.read_trace(iters, passes)
        st = eng.stats()
    finally:
        eng.close()
    assert st["status_bits"] == 0 and st["proposals"] == iters * G * T
    assert np.array_equal(terms[:, :, 8], cost)
    assert 0 < st["accepted"] < st["proposals"] and st["swaps"] > 0
    key = int(ob.lib().og_seed_key(seed.encode()))
    cfg = tb.default_config(ob.SUM_CANONICAL)
    for g in (0, 1, 31, 32, 1000, 2047, 2048, 4095):
        d = g % distinct
        o = tb.run(wl["density"][d], wl["moisture"][d], wl["temperature"][d], wl["material"][d], wl["temperatures"][d],
                   iters, f, cfg, key=key, stream_base=g * T, swap_stream_id=g)
        assert o["err"] == 0
        for fld in FIELDS:
            assert np.array_equal(trace[:, g * T:(g + 1) * T][fld], o["trace"][fld]), (g, fld)
        assert np.array_equal(swaps[:, g, :], o["swap_trace"]) and np.array_equal(cost[g], o["cost"]), g

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


def _gpu(wl, iters, f, *, seed=None, chain_u=None, swap_u=None, weights=None, chunks=None):
    eng = ChunkEngine(wl["temps"], wl["chunks"], swap_frequency=f, seed=seed, weights=weights,
                      rng_mode=nt.RNG_INJECTED if chain_u is not None else nt.RNG_PHILOX)
    try:
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
def test_trace_parity_philox(gpu_lib, chunks, T, iters, f):
    wl = wlm.chunk_workload("trace", chunks, T, index=T)
    gpu = _gpu(wl, iters, f, seed=f"chunk-{T}", chunks=[iters // 3, iters - iters // 3])
    _assert_same(wl, gpu, iters, f, seed=f"chunk-{T}")
    tr = gpu["trace"]
    assert set(np.unique(tr["kind"])) == {0, 1}
    assert (tr["flags"] & 4).any() and not (tr["flags"] & 4).all()
    assert gpu["stats"]["proposals"] == iters * chunks * T


def test_cave_search_is_exercised(gpu_lib):
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
    gpu = _gpu(wl, iters, f, chain_u=chain_u, swap_u=swap_u, weights=weights)
    _assert_same(wl, gpu, iters, f, chain_u=chain_u, swap_u=swap_u, weights=weights)
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

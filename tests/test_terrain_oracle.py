"""The voxel-chunk oracle against values derived by hand from the reference's formulas
(reference src/Terrain.Constraints.fs:12-247).  The reference's own tests hold no numeric
vector for the terrain cost, so these are this build's derivations: parity unpinned."""
import numpy as np
import pytest

import oracle_binding as ob
import terrain_binding as tb
from procgen_amd import workloads as wlm

RES, V = 32, 32 ** 3
TRAV = (V - RES * RES) * 1000.0          # the slope filter leaves the y = 0 plane reachable: 31 744 000


def _idx(x, y, z):
    return x + y * RES + z * RES * RES


def _cost(d, m=None, mat=None, mode=ob.SUM_CANONICAL):
    m = np.full(V, 0.5, np.float32) if m is None else m
    mat = np.zeros(V, np.uint8) if mat is None else mat
    return tb.cost(d, m, mat, tb.default_config(mode))


def test_traversability_is_the_constant_the_slope_filter_leaves():
    assert tb.lib().ot_traversable_reach(32) == 1024
    assert tb.lib().ot_traversable_reach(8) == 64
    assert TRAV == 31744000.0


@pytest.mark.parametrize("mode", [ob.SUM_FSHARP, ob.SUM_CANONICAL])
def test_all_air_and_all_solid(mode):
    t = _cost(np.zeros(V, np.float32), mode=mode)
    assert (t.trav, t.geo, t.overhang, t.erosion, t.aesthetics, t.caves, t.performance, t.biome) == (TRAV, 0, 0, 0, 0, 0, 0, 0)
    assert t.reached == V and t.underground_air == 17 * 32 * 32 and t.total == 8.0 * TRAV
    t = _cost(np.ones(V, np.float32), mode=mode)
    # heights 0, 0.5 .. 15.5 uniformly: variance 21.3 >= 1 -> 0; no underground air -> caves 0
    assert (t.geo, t.overhang, t.erosion, t.aesthetics, t.caves, t.performance) == (0, 0, 0, 0, 0, 0)
    assert t.reached == 1 and t.solid == V and t.total == 8.0 * TRAV


@pytest.mark.parametrize("mode", [ob.SUM_FSHARP, ob.SUM_CANONICAL])
def test_one_solid_voxel_in_air(mode):
    d = np.zeros(V, np.float32)
    d[_idx(5, 5, 5)] = 1.0
    m = np.full(V, 0.1, np.float32)
    t = _cost(d, m, mode=mode)
    assert t.geo == 6.0                 # six lattice edges with |1 - 0|^2
    assert t.overhang == 10.0           # solid over air: density * 10
    assert t.erosion == 3.0             # six exposed faces * 0.5, moisture 0.1 < 0.3
    assert t.aesthetics == 10.0         # one height: variance 0 -> 10 * (1 - 0)
    assert t.reached == V - 1 and t.caves == 500.0      # every air voxel still has a monotone air path
    assert t.surface == 1 and t.performance == 0.0
    assert t.total == 8 * TRAV + 3 * 6 + 12 * 10 + 2 * 3 + 6 * 10 + 4 * 500
    m[_idx(5, 5, 5)] = 0.3              # not < 0.3: no erosion term
    assert _cost(d, m, mode=mode).erosion == 0.0


def test_thresholds_are_strict_and_overhang_uses_the_density():
    d = np.zeros(V, np.float32)
    d[_idx(3, 4, 3)] = 0.7
    d[_idx(3, 3, 3)] = 0.5              # neither solid nor air: supports nothing, exposes nothing
    t = _cost(d)
    assert t.overhang == 0.0
    d[_idx(3, 3, 3)] = 0.49
    t = _cost(d)
    assert t.overhang == float(np.float32(0.7)) * 10.0
    assert t.solid == 1


def test_cave_search_is_directed_and_needs_air_at_both_ends():
    d = np.ones(V, np.float32)
    for x in range(6):                  # an air corridor from voxel 0 along +x, then a dead end upwards
        d[_idx(x, 0, 0)] = 0.0
    d[_idx(5, 1, 0)] = 0.0
    d[_idx(9, 0, 0)] = 0.0              # air, but not connected
    d[_idx(4, 0, 1)] = 0.0              # reachable through (4,0,0) -> +z
    t = _cost(d)
    assert t.reached == 8
    assert t.caves == (V - 8) * 1000.0 * 0.5
    d[0] = 1.0                          # voxel 0 solid: it still counts as reached, nothing else does
    assert _cost(d).reached == 1


def test_performance_budget_and_biome_classes():
    d = np.ones(V, np.float32)
    d[:] = 1.0
    ys = (np.arange(V) // RES) % RES
    d[ys % 2 == 1] = 0.0                # alternate solid / air layers: every interior solid voxel is a surface voxel
    t = _cost(d)
    interior_layers = len([y for y in range(1, 31) if y % 2 == 0])
    surface = interior_layers * 30 * 30
    assert t.surface == surface
    assert t.performance == pytest.approx((surface * 12 - 10000) * 0.001, rel=0, abs=0)
    mat = (np.arange(V) % RES >= 16).astype(np.uint8)          # two materials split at x = 16
    t = _cost(np.ones(V, np.float32), mat=mat)
    assert t.biome == 1024.0                                    # 32 x 32 crossing edges, both ends solid
    t = _cost(np.zeros(V, np.float32), mat=mat, mode=ob.SUM_FSHARP)
    seq = 0.0
    for _ in range(1024):
        seq += 0.1
    assert t.biome == seq
    tc = _cost(np.zeros(V, np.float32), mat=mat, mode=ob.SUM_CANONICAL)
    assert tc.biome == pytest.approx(102.4, rel=1e-15)


def test_canonical_and_reference_order_sums_agree_on_a_synthetic_chunk():
    wl = wlm.chunk_workload("oracle", 1, 2)
    d, m, mat = wl["density"][0, 0], wl["moisture"][0, 0], wl["material"][0]
    a = tb.cost(d, m, mat, tb.default_config(ob.SUM_FSHARP))
    b = tb.cost(d, m, mat, tb.default_config(ob.SUM_CANONICAL))
    assert a.solid == b.solid and a.reached == b.reached and a.surface == b.surface
    assert 0.2 * V < a.solid < 0.8 * V and a.underground_air > 0
    for f in ("trav", "overhang", "erosion", "caves", "performance"):
        assert getattr(a, f) == getattr(b, f)
    assert a.geo == pytest.approx(b.geo, rel=1e-12) and a.biome == pytest.approx(b.biome, rel=1e-12)
    assert a.aesthetics == pytest.approx(b.aesthetics, rel=1e-5, abs=1e-6)
    assert a.total == pytest.approx(b.total, rel=1e-12)


def test_short_run_moves_and_swaps():
    wl = wlm.chunk_workload("run", 1, 3)
    key = int(ob.lib().og_seed_key(b"terrain-run"))
    cfg = tb.default_config(ob.SUM_CANONICAL)
    r = tb.run(wl["density"][0], wl["moisture"][0], wl["temperature"][0], wl["material"][0], wl["temperatures"][0],
               40, 7, cfg, key=key)
    assert r["err"] == 0
    tr = r["trace"]
    assert set(np.unique(tr["kind"])) <= {0, 1} and (tr["kind"] == 1).any()
    assert (tr["flags"] & 4).any() and r["swap_trace"].shape == (6, 2)
    # the tracked cost is the cost of the final state
    for t in range(3):
        assert tb.cost(r["density"][t], r["moisture"][t], wl["material"][0], cfg).total == r["cost"][t]

"""Pins the CPU oracle against every known-answer value the reference holds for
the parallel-tempering path (reference tests/TestData.fs, doc-comment examples
in src/ParallelTempering.Core.fs and src/SpatialHash.fs) and against published
vectors for the third-party primitives (SHA-256, Philox4x32-10)."""
import ctypes as C
import hashlib
import math
import struct

import numpy as np
import pytest

import oracle_binding as ob

L = ob.lib()

# reference tests/TestData.fs:8-19: (current, proposed, T, P_accept, tolerance)
ACCEPTANCE_SCENARIOS = [
    (100.0, 95.0, 10.0, 1.0, 0.05),
    (50.0, 40.0, 1.0, 1.0, 0.05),
    (100.0, 105.0, 10.0, 0.606530659712633, 0.05),
    (50.0, 60.0, 10.0, 0.367879441171442, 0.05),
    (100.0, 105.0, 5.0, 0.367879441171442, 0.05),
    (50.0, 60.0, 5.0, 0.135335283236613, 0.05),
    (100.0, 105.0, 1.0, 0.006737946999085, 0.01),
    (50.0, 60.0, 1.0, 0.000045399929762, 0.001),
    (100.0, 100.0, 1.0, 1.0, 0.05),
    (100.0, 105.0, 0.1, 1.0e-217, 0.001),
]


@pytest.mark.parametrize("cur,prop,T,p,tol", ACCEPTANCE_SCENARIOS)
def test_acceptance_scenarios(cur, prop, T, p, tol):
    got = L.og_accept_probability(cur, prop, T)
    assert abs(got - p) <= tol
    # the table's values are exact to 15 digits where they are not rounded placeholders
    if p not in (1.0e-217,):
        assert got == pytest.approx(p, rel=1e-12)


def test_ladder_config1_values():
    # geometricLadder 0.5 5.0 5 (ProceduralGeneration.fs:97)
    lad = ob.geometric_ladder(0.5, 5.0, 5)
    assert lad[0] == 5.0
    assert lad == pytest.approx([5.0, 2.8117066259517456, 1.5811388300841898, 0.8891397050194614, 0.5], rel=1e-14)
    # the code at ParallelTempering.Core.fs:92-95, not the doc comment at :87
    assert ob.geometric_ladder(0.5, 5.0, 4) == pytest.approx([5.0, 2.320794416806389, 1.0772173450159417, 0.5], rel=1e-14)


@pytest.mark.parametrize("tmin,tmax,n", [(0.5, 5.0, 4), (1.0, 10.0, 8), (0.1, 100.0, 16)])  # TestData.fs:21-25
def test_ladder_properties(tmin, tmax, n):
    lad = ob.geometric_ladder(tmin, tmax, n)
    assert len(lad) == n
    assert abs(lad[0] - tmax) < 1e-3 and abs(lad[-1] - tmin) < 1e-3      # ParallelTemperingTests.fs:22-29
    ratios = lad[:-1] / lad[1:]
    assert np.all(np.abs(ratios - ratios.mean()) < 0.01)                 # :32-43
    assert np.all(np.diff(lad) < 0)                                      # :17-20


def test_ladder_degenerate():
    assert list(ob.geometric_ladder(0.5, 5.0, 1)) == [5.0]               # ParallelTemperingTests.fs:46-50
    assert list(ob.geometric_ladder(-1.0, 5.0, 4)) == [5.0]
    assert list(ob.geometric_ladder(6.0, 5.0, 4)) == [5.0]


def test_swap_probability():
    # ParallelTemperingTests.fs:53-69
    assert L.og_swap_probability(100.0, 1.0, 50.0, 2.0) == pytest.approx(math.exp((50.0 - 100.0) * (1.0 - 0.5)), rel=1e-15)
    # worked example ParallelTempering.Core.fs:100-102: exp(40), clamps to 1 in attemptSwap
    assert L.og_swap_probability(100.0, 5.0, 50.0, 1.0) == pytest.approx(math.exp(40.0), rel=1e-15)
    assert L.og_swap_probability(100.0, 0.0, 50.0, 2.0) == 0.0           # :72-79


def test_swap_frequency():
    f = L.og_swap_frequency
    assert [f(20), f(10**4), f(10**3), f(10**6)] == [8, 200, 63, 2000]
    assert f(1) >= 5 and f(5) < f(50) < f(500)                           # ParallelTemperingTests.fs:172-184


def test_connectivity_examples():
    # "5 nodes, 3 unreachable -> 3000" ParallelTempering.Core.fs:247
    n = 5
    pos = np.zeros((n, 3), np.float32); pos[:, 0] = np.arange(n) * 10
    cfg = ob.default_config(ob.SUM_FSHARP, ob.ALGO_FAITHFUL)
    t = ob.cost(pos, [[0, 1]], ob.default_kinds(n), np.zeros(n), cfg)
    assert t.reach0 == 2 and t.conn == 3000.0
    # fully connected with only i<j edges -> 0 (ConstraintsTests.fs:21-24, Builders.fs:347-353)
    edges = [[i, j] for i in range(n) for j in range(i + 1, n)]
    t = ob.cost(pos, edges, ob.default_kinds(n), np.zeros(n), cfg)
    assert t.conn == 0.0 and t.endpoint == 0.0
    # BFS is directed: b->a edges do not reach (Execution.fs:205)
    assert ob.count_reachable(3, [[1, 0], [2, 1]], 0) == 1
    assert ob.count_reachable(3, [[0, 1], [1, 2]], 0) == 3
    assert ob.count_reachable(3, [[0, 1]], 5) == 0


def test_endpoint_counts_vs_index_quirk():
    # ProceduralGeneration.fs:53 compares reachable COUNT with terminal INDEX
    n = 4
    pos = np.zeros((n, 3), np.float32); pos[:, 0] = np.arange(n) * 10
    kind = np.array([ob.ORIGIN, ob.TERMINAL, ob.STANDARD, ob.STANDARD], np.uint8)
    cfg = ob.default_config(ob.SUM_FSHARP, ob.ALGO_FAITHFUL)
    # origin reaches node 2 only (count 2 > terminal index 1) although the terminal is unreachable
    assert ob.cost(pos, [[0, 2]], kind, np.zeros(n), cfg).endpoint == 0.0
    assert ob.cost(pos, [], kind, np.zeros(n), cfg).endpoint == 10000.0
    kind2 = np.full(n, ob.STANDARD, np.uint8)
    assert ob.cost(pos, [[0, 1]], kind2, np.zeros(n), cfg).endpoint == 10000.0


def _coll(p1, p2, mind):
    cfg = ob.default_config(ob.SUM_FSHARP, ob.ALGO_FAITHFUL)
    cfg.min_dist = mind
    cfg.w[:] = [0, 0, 0, 1, 0, 0]
    pos = np.array([p1, p2], np.float32)
    return ob.cost(pos, [], ob.default_kinds(2), np.zeros(2), cfg)


def test_collision_cases():
    # TestData.fs:35-44
    assert _coll([0, 0, 0], [10, 0, 0], 1.0).npairs == 0
    assert _coll([0, 0, 0], [0.5, 0, 0], 1.0).npairs == 1
    assert _coll([0, 0, 0], [1.0, 0, 0], 1.0).npairs == 0      # strict < at dist == minDistance
    assert _coll([0, 0, 0], [1, 1, 1], 2.0).npairs == 1
    # "distance 0.3, minDistance 0.5 -> overlap 0.2": code multiplies by 100 (SpatialHash.fs:190)
    t = _coll([0, 0, 0], [0.3, 0, 0], 0.5)
    d = float(np.float32(0.3))
    assert t.coll == (0.5 - d) * (0.5 - d) * 100.0
    assert t.coll == pytest.approx(4.0, rel=1e-6)
    # coincident points are NOT collisions: dist > 0.001 guard (SpatialHash.fs:98,143)
    assert _coll([1, 1, 1], [1, 1, 1], 0.5).npairs == 0


def test_sha256_and_hashrng():
    for msg in [b"", b"abc", b"a" * 55, b"a" * 56, b"a" * 64, b"procgen" * 40]:
        out = C.create_string_buffer(32)
        L.og_sha256(msg, len(msg), out)
        assert out.raw == hashlib.sha256(msg).digest()
    for seed, idx in [("seed", 0), ("abc", 17), ("x" * 30, 123456), ("s", -5)]:
        dg = hashlib.sha256((seed + str(idx)).encode()).digest()
        want = struct.unpack("<Q", dg[:8])[0] / 18446744073709551616.0
        assert L.og_hashrng_next_float(seed.encode(), idx) == want
        assert L.og_hashrng_next_int(seed.encode(), idx, 1000) == int(want * 1000)
    assert L.og_seed_key(b"abc") == struct.unpack("<Q", hashlib.sha256(b"abc").digest()[:8])[0]


def _hash_coords_py(base, x, y, z):
    M = (1 << 64) - 1
    h = base
    h ^= ((x & M) * 0xff51afd7ed558ccd) & M; h ^= h >> 33
    h ^= ((y & M) * 0xc4ceb9fe1a85ec53) & M; h ^= h >> 33
    h ^= ((z & M) * 0xff51afd7ed558ccd) & M; h ^= h >> 33
    return h


def test_hash_coordinates():
    for args in [(0, 0, 0, 0), (12345, 1, 2, 3), (2**63 + 5, -1, -2, 7), (42, -100, 100, -3)]:
        assert L.og_hash_coordinates(*args) == _hash_coords_py(*args)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    kats = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kats:
        c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
        L.og_philox4x32_10(c, k, o)
        assert tuple(o) == want


def test_philox_stream_layout():
    u = ob.philox_stream(0x0123456789abcdef, 3, 0, 6)
    assert np.all((u >= 0) & (u < 1))
    c = (C.c_uint32 * 4)(1, 0, 3, 0); k = (C.c_uint32 * 2)(0x89abcdef, 0x01234567); o = (C.c_uint32 * 4)()
    L.og_philox4x32_10(c, k, o)
    assert u[2] == ((o[1] >> 5) * 67108864.0 + (o[0] >> 6)) / 9007199254740992.0
    assert u[3] == ((o[3] >> 5) * 67108864.0 + (o[2] >> 6)) / 9007199254740992.0


def test_uniform_sphere_matches_formula():
    # ProceduralGeneration.fs:17-33
    n = 20
    pos = ob.uniform_sphere(n)
    ga = math.pi * (3.0 - math.sqrt(5.0))
    for i in range(n):
        y = 1.0 - (i / (n - 1)) * 2.0
        r = math.sqrt(1.0 - y * y)
        want = np.array([r * math.cos(i * ga), y, r * math.sin(i * ga)]).astype(np.float32)
        assert np.array_equal(pos[i], want)
    assert np.all(ob.uniform_sphere(1) == 0)

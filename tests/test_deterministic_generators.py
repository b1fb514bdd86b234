"""The seeded workload generators of the reference (src/Deterministic.fs:11-135): the host-side mirror
`procgen_amd.deterministic` against the C oracle (independent code) and against values derived by hand
from the reference's formulas.  The reference holds no recorded output of these generators, and .NET's
Math.Cos / Math.Sin are not available here: beyond the formulas, parity is unpinned (fp64 trigonometry
from two C libraries can differ in the last bit before the rounding to fp32)."""
import hashlib
import math
import struct

import numpy as np
import pytest

import oracle_binding as ob
from procgen_amd import deterministic as det
from procgen_amd import native as nt


def _u(seed, index):
    return struct.unpack("<Q", hashlib.sha256((seed + str(index)).encode()).digest()[:8])[0] / 2.0 ** 64


def test_hashrng_mirror():
    for seed, idx in [("seed", 0), ("abc", 17), ("x" * 30, 123456), ("s", -5)]:
        assert det.HashRNG.nextFloat(seed, idx) == _u(seed, idx) == ob.lib().og_hashrng_next_float(seed.encode(), idx)
        assert det.HashRNG.nextInt(seed, idx, 1000) == ob.lib().og_hashrng_next_int(seed.encode(), idx, 1000)
    v = det.HashRNG.nextVector3("seed", 4)
    assert v.dtype == np.float32
    assert [float(c) for c in v] == [float(np.float32(_u("seed", 12 + k) * 2.0 - 1.0)) for k in range(3)]


def test_fibonacci_sphere_by_hand():
    # Deterministic.fs:48-49: degenerate counts
    assert det.DeterministicDistribution.fibonacciSphere("s", 0, 0.1).shape == (0, 3)
    assert np.array_equal(det.DeterministicDistribution.fibonacciSphere("s", 1, 0.1), np.zeros((1, 3), np.float32))
    p = det.DeterministicDistribution.fibonacciSphere("s", 5, 0.0)
    # i = 0: y = 1, radius = 0 -> the north pole; i = 4: y = -1 -> the south pole; i = 2: the equator
    assert p[0].tolist() == [0.0, 1.0, 0.0]
    assert p[4, 1] == -1.0 and abs(p[4, 0]) == 0.0 and abs(p[4, 2]) == 0.0
    theta = (1.0 + math.sqrt(5.0)) / 2.0 * 2.0 * 2.0 * math.pi
    assert p[2].tolist() == [float(np.float32(math.cos(theta))), 0.0, float(np.float32(math.sin(theta)))]
    # unit sphere
    big = det.DeterministicDistribution.fibonacciSphere("s", 257, 0.0)
    assert np.allclose(np.linalg.norm(big.astype(np.float64), axis=1), 1.0, atol=1e-6)
    # the perturbation is `noise * float32 perturbation` added in fp32, noise in [-1, 1]
    q = det.DeterministicDistribution.fibonacciSphere("s", 257, 0.1)
    for i in (0, 3, 256):
        noise = det.HashRNG.nextVector3("s", i)
        assert np.array_equal(q[i], big[i] + noise * np.float32(0.1))
    assert np.max(np.abs(q - big)) <= 0.1 + 1e-6


@pytest.mark.parametrize("n,pert", [(2, 0.1), (37, 0.0), (400, 0.1), (1500, 0.25)])
def test_fibonacci_sphere_matches_oracle(n, pert):
    a = det.DeterministicDistribution.fibonacciSphere("seed-%d" % n, n, pert)
    b = ob.fibonacci_sphere("seed-%d" % n, n, pert)
    assert a.tobytes() == b.tobytes()


def test_seeded_uniform_sphere():
    a = det.DeterministicDistribution.uniformSphere("ball", 300, 2.5)
    b = ob.seeded_uniform_sphere("ball", 300, 2.5)
    assert a.tobytes() == b.tobytes()
    r = np.linalg.norm(a.astype(np.float64), axis=1)
    assert r.max() <= 2.5 + 1e-6 and r.min() >= 0.0
    # by hand, point 1 (Deterministic.fs:74-86)
    u, v, w = _u("ball", 3), _u("ball", 4), _u("ball", 5)
    th, ph, rr = 2.0 * math.pi * u, math.acos(2.0 * v - 1.0), math.pow(w, 1.0 / 3.0) * 2.5
    want = [np.float32(rr * math.sin(ph) * math.cos(th)), np.float32(rr * math.sin(ph) * math.sin(th)), np.float32(rr * math.cos(ph))]
    assert a[1].tolist() == [float(x) for x in want]


@pytest.mark.parametrize("n", [1, 2, 3, 60, 700])
def test_create_graph_and_edges_match_oracle(n):
    seed = "graph-%d" % n
    g = det.DeterministicGeneration.createGraph(seed, n, 0.1)
    pos, kind, value = ob.create_graph(seed, n, 0.1)
    assert g["pos"].tobytes() == pos.tobytes()
    assert np.array_equal(g["kind"], kind) and np.array_equal(g["value"], value)
    assert kind[0] == nt.NODE_ORIGIN
    if n > 1:
        assert kind[n - 1] == nt.NODE_TERMINAL
    for i in range(1, n - 1):
        assert value[i] == _u(seed, i + 1000)
    g2 = det.DeterministicGeneration.generateEdges(seed, g, 3.0)
    e, w = ob.generate_edges(seed, pos, 3.0)
    assert np.array_equal(g2["edges"], e) and np.array_equal(g2["weights"], w)
    assert len(e) <= int(n * 3.0 / 2.0)
    if n >= 60:
        assert len(e) == int(n * 3.0 / 2.0)                      # the target is reached within the first rows
        assert np.all(e[:, 0] < e[:, 1])
        # list order = reverse of discovery order: (i, j) pairs strictly decreasing in row-major order
        key = e[:, 0].astype(np.int64) * n + e[:, 1]
        assert np.all(np.diff(key) < 0)


def test_generate_edges_by_hand():
    """Small graph, every pair re-derived from the formula (Deterministic.fs:125-133)."""
    seed, n = "hand", 12
    g = det.DeterministicGeneration.createGraph(seed, n, 0.1)
    pos = g["pos"]
    want = []
    target = int(n * 3.0 / 2.0)
    for i in range(n):
        for j in range(i + 1, n):
            d = pos[i] - pos[j]
            s = np.float32(np.float32(d[0] * d[0]) + np.float32(d[1] * d[1]))
            s = np.float32(s + np.float32(d[2] * d[2]))
            dist = float(np.sqrt(s))
            if _u(seed, i * 1000 + j) < 1.0 / (1.0 + dist * dist) and len(want) < target:
                want.append((i, j))
    got = det.DeterministicGeneration.generateEdges(seed, g, 3.0)["edges"]
    assert got.tolist() == [list(p) for p in want[::-1]]


def test_deterministic_workload_shape():
    w = det.deterministic_workload("job", 90, 4)
    assert w["pos"].shape == (1, 4, 90, 3) and w["edges"].shape[:2] == (1, 4)
    assert w["n_edges"] == 135 and np.all(w["ecount"] == 135)
    for t in range(1, 4):
        assert np.array_equal(w["pos"][0, t], w["pos"][0, 0]) and np.array_equal(w["edges"][0, t], w["edges"][0, 0])
    assert np.array_equal(w["temperatures"][0], nt.geometric_ladder(0.5, 5.0, 4))

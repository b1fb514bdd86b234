"""Seeded synthetic inputs for the parallel-tempering engine.

All generators are deterministic functions of a seed string: the generator key
is the little-endian u64 of SHA-256("procgen-bench/<name>/<index>")[0..8],
the same seed-to-key rule the engine uses for its Philox streams.
"""
import hashlib
import math
import struct

import numpy as np

from . import native as nt


def _rng(name, index=0):
    key = struct.unpack("<Q", hashlib.sha256(f"procgen-bench/{name}/{index}".encode()).digest()[:8])[0]
    return np.random.Generator(np.random.Philox(key=key))


def uniform_sphere(n):
    """Golden-angle unit sphere (reference src/ProceduralGeneration.fs:17-33)."""
    if n <= 0:
        return np.zeros((0, 3), np.float32)
    if n == 1:
        return np.zeros((1, 3), np.float32)
    golden = math.pi * (3.0 - math.sqrt(5.0))
    i = np.arange(n, dtype=np.float64)
    y = 1.0 - (i / float(n - 1)) * 2.0
    radius = np.sqrt(1.0 - y * y)
    theta = i * golden
    return np.stack([radius * np.cos(theta), y, radius * np.sin(theta)], axis=1).astype(np.float32)


def box_positions(n, rng):
    """Uniform in a cube of side 0.5*(2n)^(1/3): about half a node per 0.5-wide cell."""
    side = 0.5 * (2.0 * n) ** (1.0 / 3.0)
    return ((rng.random((n, 3)) - 0.5) * side).astype(np.float32)


def path_plus_random_edges(n, n_edges, rng):
    """Directed edges: the spanning path i -> i+1 plus random extra pairs with no
    duplicate in either direction (what Mutations.addEdge would accept,
    reference src/ParallelTempering.Core.fs:55).  Returned in LIST order
    (index 0 = list head); the order is a seeded shuffle."""
    n_edges = min(max(n_edges, n - 1), n * (n - 1) // 2)
    seen = set()
    edges = []
    for i in range(n - 1):
        edges.append((i, i + 1)); seen.add((i, i + 1))
    while len(edges) < n_edges and n >= 2:
        need = n_edges - len(edges)
        a = rng.integers(0, n, size=need * 2)
        b = rng.integers(0, n, size=need * 2)
        for x, y in zip(a.tolist(), b.tolist()):
            if x == y or (x, y) in seen or (y, x) in seen:
                continue
            seen.add((x, y)); edges.append((x, y))
            if len(edges) == n_edges:
                break
    e = np.array(edges, np.int32).reshape(-1, 2)
    return e[rng.permutation(len(e))]


def node_attributes(n, rng):
    """createGraph's typing (reference src/ProceduralGeneration.fs:37-41): node 0 Origin,
    n-1 Terminal, the rest Standard u."""
    kind = np.full(n, nt.NODE_STANDARD, np.uint8)
    if n > 0:
        kind[0] = nt.NODE_ORIGIN
    if n > 1:
        kind[n - 1] = nt.NODE_TERMINAL
    value = rng.random(n)
    return kind, value


def graph_workload(name, n, temps, groups=1, *, layout="box", avg_degree=3.0, edge_headroom=1.25,
                   t_min=0.5, t_max=5.0, index=0):
    """One synthetic PT job: `groups` independent ladders of `temps` replicas over n-node graphs.

    Every replica of a ladder starts from the same graph (as the reference's
    generators do, src/Deterministic.fs:166).  E = n*avg_degree/2 directed edges
    (src/Deterministic.fs:120,140).  Returns a dict ready for GraphEngine.upload.
    """
    n_edges = max(int(n * avg_degree / 2.0), n - 1)
    ecap = int(n_edges * edge_headroom) + 16
    pos = np.zeros((groups, temps, n, 3), np.float32)
    edges = np.zeros((groups, temps, ecap, 2), np.int32)
    ecount = np.zeros((groups, temps), np.int32)
    kind = np.zeros((groups, n), np.uint8)
    value = np.zeros((groups, n), np.float64)
    for g in range(groups):
        rng = _rng(name, index * 1000003 + g)
        p = uniform_sphere(n) if layout == "sphere" else box_positions(n, rng)
        e = path_plus_random_edges(n, n_edges, rng)
        k, v = node_attributes(n, rng)
        pos[g, :] = p
        edges[g, :, :len(e)] = e
        ecount[g, :] = len(e)
        kind[g], value[g] = k, v
    ladder = nt.geometric_ladder(t_min, t_max, temps)
    temperatures = np.tile(ladder, (groups, 1))
    return dict(n=n, temps=temps, groups=groups, ecap=ecap, pos=pos, edges=edges, ecount=ecount,
                kind=kind, value=value, temperatures=temperatures, n_edges=n_edges)


def algorithmic_bytes_per_proposal(n, e):
    """Bytes one full cost re-evaluation must touch: positions 12N, edge endpoints 8E,
    node values 4N, CSR 4(N+1)+4E, visited bitmap N/4 (read+write)."""
    return 12 * n + 8 * e + 4 * n + 4 * (n + 1) + 4 * e + n / 4.0


def algorithmic_flops_per_proposal(n, e, mean_candidates=13.5):
    """VALU lane-ops of one full cost re-evaluation: 11E edge length, 30N distribution,
    (27 + 10 k)N collision with k candidates per node, 2E transitions."""
    return 11 * e + 30 * n + (27 + 10 * mean_candidates) * n + 2 * e

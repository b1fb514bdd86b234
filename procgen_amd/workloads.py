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


# ---- voxel chunks ---------------------------------------------------------------------------------
MAT_AIR, MAT_STONE, MAT_SOIL, MAT_SAND, MAT_SNOW, MAT_ICE, MAT_GRASS, MAT_WATER = range(8)    # reference src/Voxel.Core.fs:79-87


def determine_material(density, temperature, moisture, elevation):
    """Material.determineMaterial (reference src/Voxel.Core.fs:225-235), vectorised."""
    d, t, m, e = (np.asarray(a, np.float32) for a in (density, temperature, moisture, elevation))
    out = np.where(m > np.float32(0.5), MAT_GRASS, MAT_SOIL)
    out = np.where(e > np.float32(0.8), MAT_STONE, out)
    out = np.where(e < np.float32(0.3), np.where(m > np.float32(0.6), MAT_SOIL, MAT_SAND), out)
    out = np.where(t < np.float32(0.0), np.where(m > np.float32(0.7), MAT_ICE, MAT_SNOW), out)
    out = np.where(d < np.float32(0.5), MAT_AIR, out)
    return out.astype(np.uint8)


def chunk_workload(name, chunks, temps=6, *, res=32, t_min=0.1, t_max=5.0, index=0, air_pockets=0.03):
    """Synthetic voxel chunks: density 1 below a seeded value-noise height field and 0 above, a few
    per cent of random air pockets below mid-height, moisture U(0,1), voxel temperature U(-10,30),
    materials by determineMaterial.  Every replica of a chunk's ladder starts from the same chunk
    (reference src/Terrain.Generation.fs:181-182); ladder = geometricLadder 0.1 5.0 6 (:18-23).
    Arrays: density/moisture/temperature [chunks, temps, V], material [chunks, V]."""
    V = res ** 3
    dens = np.zeros((chunks, temps, V), np.float32)
    moist = np.zeros((chunks, temps, V), np.float32)
    temp = np.zeros((chunks, temps, V), np.float32)
    mat = np.zeros((chunks, V), np.uint8)
    v = np.arange(V)
    x, y, z = v % res, (v // res) % res, v // (res * res)
    for g in range(chunks):
        rng = _rng("chunk-" + name, index * 1000003 + g)
        coarse = rng.uniform(0.25 * res, 0.75 * res, size=(5, 5))
        gx = x / (res - 1) * 4.0
        gz = z / (res - 1) * 4.0
        x0, z0 = np.minimum(gx.astype(int), 3), np.minimum(gz.astype(int), 3)
        fx, fz = gx - x0, gz - z0
        h = (coarse[x0, z0] * (1 - fx) * (1 - fz) + coarse[x0 + 1, z0] * fx * (1 - fz) +
             coarse[x0, z0 + 1] * (1 - fx) * fz + coarse[x0 + 1, z0 + 1] * fx * fz)
        d = (y < h).astype(np.float32)
        pockets = (rng.random(V) < air_pockets) & (y <= res // 2)
        d[pockets] = 0.0
        m = rng.random(V).astype(np.float32)
        t = rng.uniform(-10.0, 30.0, V).astype(np.float32)
        dens[g, :], moist[g, :], temp[g, :] = d, m, t
        mat[g] = determine_material(d, t, m, (y * np.float32(0.5)) / np.float32(100.0))
    ladder = nt.geometric_ladder(t_min, t_max, temps)
    return dict(chunks=chunks, temps=temps, res=res, V=V, density=dens, moisture=moist, temperature=temp, material=mat,
                temperatures=np.tile(ladder, (chunks, 1)))


def chunk_algorithmic_bytes_per_proposal(V):
    """Bytes one full terrain-cost evaluation must touch: density and moisture f32, material u8,
    visited bitmap read+write."""
    return 9 * V + V / 4.0


def algorithmic_bytes_per_proposal(n, e):
    """Bytes one full cost re-evaluation must touch: positions 12N, edge endpoints 8E,
    node values 4N, CSR 4(N+1)+4E, visited bitmap N/4 (read+write)."""
    return 12 * n + 8 * e + 4 * n + 4 * (n + 1) + 4 * e + n / 4.0


def algorithmic_flops_per_proposal(n, e, mean_candidates=13.5):
    """VALU lane-ops of one full cost re-evaluation: 11E edge length, 30N distribution,
    (27 + 10 k)N collision with k candidates per node, 2E transitions."""
    return 11 * e + 30 * n + (27 + 10 * mean_candidates) * n + 2 * e

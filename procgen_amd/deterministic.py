"""Host-side mirror of the reference's seeded workload generators (src/Deterministic.fs).

Same module and function names and the same formulas as the F# file: a job described by a seed
string builds its starting graph here the way `DeterministicGeneration.createGraph` /
`generateEdges` build it there.  Everything is a pure function of the seed string:
`HashRNG` hashes utf8(seed + decimal(index)) with SHA-256 and reads the first eight bytes.

How far "the same" goes.  `HashRNG` and `hashCoordinates` are integer and SHA-256 arithmetic:
exact.  The point generators go through fp64 `cos` / `sin` (arguments up to 1e7 rad), `acos`
and `pow`, taken here from the C library and there from .NET's `Math.*`; a one-ulp difference
between the two can survive the rounding to fp32 in rare cases and then move a coordinate by one
fp32 ulp or flip a `threshold < probability` edge decision.  The reference holds no output of
these generators to check against, so their results are parity unpinned: equal to this
repository's own restatement (oracle/pt_oracle.c), not claimed bit-identical to a .NET run.

The optimisation loop of `DeterministicGeneration.generate` (src/Deterministic.fs:143-178)
is not mirrored: its mutation seeds itself from `graph.GetHashCode()`, a .NET object
identity that differs from run to run; the engine's addressed Philox draws take its place.
"""
import hashlib
import math

import numpy as np

from . import native as nt

_F32 = np.float32


class HashRNG:
    """src/Deterministic.fs:11-40."""

    @staticmethod
    def generateBytes(seed, index):
        """SHA-256 of utf8(seed + index.ToString()) (src/Deterministic.fs:14-17)."""
        return hashlib.sha256((seed + str(int(index))).encode("utf-8")).digest()

    @staticmethod
    def bytesToFloat(b):
        """First 8 bytes as a little-endian u64 over `float UInt64.MaxValue`, which is 2^64 as a
        double, so the result lies in [0, 1] with 1.0 possible (src/Deterministic.fs:20-22)."""
        return float(int.from_bytes(b[:8], "little")) / 18446744073709551616.0

    @staticmethod
    def nextFloat(seed, index):
        return HashRNG.bytesToFloat(HashRNG.generateBytes(seed, index))

    @staticmethod
    def nextVector3(seed, index):
        """Draws 3i, 3i+1, 3i+2 mapped to [-1, 1] in fp64, rounded to fp32 (src/Deterministic.fs:30-35)."""
        return np.array([_F32(HashRNG.nextFloat(seed, index * 3 + k) * 2.0 - 1.0) for k in range(3)], _F32)

    @staticmethod
    def nextInt(seed, index, max_):
        return int(HashRNG.nextFloat(seed, index) * float(max_))


class DeterministicDistribution:
    """src/Deterministic.fs:43-87."""

    phi = (1.0 + math.sqrt(5.0)) / 2.0

    @staticmethod
    def fibonacciSphere(seed, count, perturbation):
        """src/Deterministic.fs:47-68: y from 1 to -1, theta = phi * i * 2 * pi (multiplied in that
        order), fp64 trigonometry rounded to fp32, then `noise * float32 perturbation` added in fp32."""
        if count <= 0:
            return np.zeros((0, 3), _F32)
        if count == 1:
            return np.zeros((1, 3), _F32)
        out = np.zeros((count, 3), _F32)
        s = _F32(perturbation)
        for i in range(count):
            y = 1.0 - (float(i) / float(count - 1)) * 2.0
            radius = math.sqrt(1.0 - y * y)
            theta = DeterministicDistribution.phi * float(i) * 2.0 * math.pi
            base = np.array([_F32(radius * math.cos(theta)), _F32(y), _F32(radius * math.sin(theta))], _F32)
            if perturbation > 0.0:
                base = base + HashRNG.nextVector3(seed, i) * s
            out[i] = base
        return out

    @staticmethod
    def uniformSphere(seed, count, radius):
        """Points in a ball (src/Deterministic.fs:71-87)."""
        out = np.zeros((max(count, 0), 3), _F32)
        for i in range(count):
            u = HashRNG.nextFloat(seed, i * 3)
            v = HashRNG.nextFloat(seed, i * 3 + 1)
            w = HashRNG.nextFloat(seed, i * 3 + 2)
            theta = 2.0 * math.pi * u
            ph = math.acos(2.0 * v - 1.0)
            r = math.pow(w, 1.0 / 3.0) * radius
            out[i] = (_F32(r * math.sin(ph) * math.cos(theta)), _F32(r * math.sin(ph) * math.sin(theta)),
                      _F32(r * math.cos(ph)))
        return out


def _distance(p, q):
    """System.Numerics.Vector3.Distance in fp32: sqrt((dx*dx + dy*dy) + dz*dz)."""
    d = p - q
    s = _F32(_F32(d[0] * d[0]) + _F32(d[1] * d[1]))
    s = _F32(s + _F32(d[2] * d[2]))
    return np.sqrt(s, dtype=_F32)


class DeterministicGeneration:
    """src/Deterministic.fs:90-135 (`hashCoordinates` is `native.hash_coordinates`, from the library)."""

    @staticmethod
    def createGraph(seed, nodeCount, perturbation):
        """Positions on the perturbed fibonacci sphere; node 0 Origin, the last Terminal, the others
        Standard (nextFloat seed (i + 1000)); no edges (src/Deterministic.fs:105-116).
        Returns dict(pos [n,3] f32, kind [n] u8, value [n] f64, edges [0,2] i32)."""
        pos = DeterministicDistribution.fibonacciSphere(seed, nodeCount, perturbation)
        kind = np.full(nodeCount, nt.NODE_STANDARD, np.uint8)
        value = np.zeros(nodeCount, np.float64)
        for i in range(nodeCount):
            if i == 0:
                kind[i] = nt.NODE_ORIGIN
            elif i == nodeCount - 1:
                kind[i] = nt.NODE_TERMINAL
            else:
                value[i] = HashRNG.nextFloat(seed, i + 1000)
        return dict(pos=pos, kind=kind, value=value, edges=np.zeros((0, 2), np.int32),
                    weights=np.zeros(0, np.float64))

    @staticmethod
    def generateEdges(seed, graph, avgDegree):
        """Pairs i < j in row order, kept when nextFloat seed (i*1000 + j) < 1 / (1 + d^2), until
        int(n * avgDegree / 2) edges exist (src/Deterministic.fs:119-135).  The reference hashes the
        remaining pairs too, without effect; this stops at the target.  Each edge is consed onto the
        list, so `edges` (LIST order, row 0 = head) is the reverse of the order of discovery."""
        pos = graph["pos"]
        n = len(pos)
        target = int(float(n) * avgDegree / 2.0)
        found, weights = [], []
        for i in range(n):
            if len(found) >= target:
                break
            for j in range(i + 1, n):
                if len(found) >= target:
                    break
                dist = _distance(pos[i], pos[j])
                probability = 1.0 / (1.0 + float(dist) * float(dist))
                index = (i * 1000 + j) & 0xFFFFFFFF
                if index >= 0x80000000:
                    index -= 0x100000000              # the reference's index is a wrapping 32-bit int
                if HashRNG.nextFloat(seed, index) < probability:
                    found.append((i, j))
                    weights.append(float(dist))
        out = dict(graph)
        out["edges"] = np.array(found[::-1], np.int32).reshape(-1, 2)
        out["weights"] = np.array(weights[::-1], np.float64)
        return out


def deterministic_workload(seed, n, temps, *, perturbation=0.1, avg_degree=3.0, edge_headroom=1.25,
                           t_min=0.5, t_max=5.0):
    """The starting point of `DeterministicGeneration.generate` (src/Deterministic.fs:138-140,166):
    createGraph seed n 0.1, generateEdges 3.0, every replica of the ladder starting from that graph.
    Returns a dict shaped like `workloads.graph_workload` (one ladder)."""
    g = DeterministicGeneration.generateEdges(seed, DeterministicGeneration.createGraph(seed, n, perturbation),
                                              avg_degree)
    e = g["edges"]
    ecap = int(max(len(e), n - 1) * edge_headroom) + 16
    pos = np.zeros((1, temps, n, 3), np.float32)
    edges = np.zeros((1, temps, ecap, 2), np.int32)
    ecount = np.full((1, temps), len(e), np.int32)
    pos[0, :] = g["pos"]
    edges[0, :, :len(e)] = e
    ladder = nt.geometric_ladder(t_min, t_max, temps)
    return dict(n=n, temps=temps, groups=1, ecap=ecap, pos=pos, edges=edges, ecount=ecount,
                kind=g["kind"][None, :].copy(), value=g["value"][None, :].copy(),
                temperatures=ladder[None, :].copy(), n_edges=len(e))

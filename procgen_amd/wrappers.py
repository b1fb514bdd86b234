"""Host-side mirror of the reference's GPU dispatch layer for the parallel-tempering path
(reference src/GPU.Wrappers.fs), over libgpu_compute.so.

In a deployment the F# host keeps its own GPU.Wrappers.fs / GPU.Compute.fs; this module
restates the same call sequence (same names, argument meaning and error behaviour) so the
drop-in entry point can be exercised exactly the way that host drives it:

  Wrappers.available()            reference GPU.Wrappers.fs:109-120
  Marshaling.*                    reference GPU.Wrappers.fs:191-244
  GPUTempering.validateGraph      reference GPU.Wrappers.fs:250-258
  GPUTempering.optimize           reference GPU.Wrappers.fs:271-344  (one native call per iteration)
  GPUCollision.detect             reference GPU.Wrappers.fs:347-416
  Dispatch.useGPU / optimize      reference GPU.Wrappers.fs:689-724
  Dispatch.optimizeUntilConvergence  reference GPU.Wrappers.fs:728-753

plus `optimize_batched`, the one-call replacement the reference's glue could switch to.
Results are (ok, value) / (err, CudaError) pairs standing in for F#'s Result.
"""
import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import native as nt
from .engine import GraphEngine


# ---- reference GPU.Wrappers.fs:12-18 ------------------------------------------------------------
@dataclass
class CudaError:
    kind: str           # DeviceNotAvailable | InvalidArgument | BufferOverflow | NativeException
    message: str = ""


def Ok(v):
    return ("ok", v)


def Error(kind, message=""):
    return ("error", CudaError(kind, message))


@dataclass
class SpatialGraph:
    """reference src/ParallelTempering.Core.fs:8-12; Edges is a list with the newest edge first."""
    Nodes: list
    Positions: np.ndarray                     # [n, 3] float32
    Edges: List[Tuple[int, int, float]] = field(default_factory=list)


class Wrappers:
    _gpu_threshold = 10000                    # reference GPU.Wrappers.fs:100

    @staticmethod
    def available() -> bool:
        """The probe call with empty arrays and zero counts (reference GPU.Wrappers.fs:109-120);
        False when the library cannot be loaded or sees no device."""
        try:
            lib = nt.load()
            lib.parallel_tempering_step(None, None, None, None, 0, 0, 0)
            return lib.pt_last_error() == 0             # PT_ERR_NO_DEVICE: the process has no usable GPU
        except Exception:
            return False

    @classmethod
    def getGPUThreshold(cls) -> int:
        return cls._gpu_threshold

    @classmethod
    def updateGPUThreshold(cls, new_threshold: int) -> None:
        cls._gpu_threshold = max(1000, min(1000000, int(new_threshold)))     # reference :126-127

    @classmethod
    def shouldUseGPU(cls, problem_size: int) -> bool:
        return problem_size >= cls._gpu_threshold and cls.available()       # reference :184-188


class Marshaling:
    @staticmethod
    def flattenVec3(vectors) -> np.ndarray:
        """Vector3 array -> float (System.Double) array, reference GPU.Wrappers.fs:198-199."""
        return np.asarray(vectors, np.float32).reshape(-1, 3).astype(np.float64).reshape(-1)

    @staticmethod
    def unflattenVec3(floats) -> np.ndarray:
        return np.asarray(floats, np.float64).reshape(-1, 3).astype(np.float32)

    @staticmethod
    def flattenEdges(edges) -> np.ndarray:
        """(a, b, w) list -> int array a0,b0,a1,b1,... in list order, reference :208-209."""
        return np.array([[a, b] for (a, b, *_) in edges], np.int32).reshape(-1)

    @staticmethod
    def toVector3Array(floats):
        floats = np.asarray(floats, np.float64)
        if floats.size % 3 != 0:
            return Error("InvalidArgument", f"Float array length {floats.size} not divisible by 3")
        return Ok(Marshaling.unflattenVec3(floats))


class GPUTempering:
    @staticmethod
    def validateGraph(graph: SpatialGraph):
        """reference GPU.Wrappers.fs:250-258"""
        n = len(graph.Nodes)
        if len(graph.Positions) != n:
            return Error("InvalidArgument", f"Position count {len(graph.Positions)} != Node count {n}")
        if any(a < 0 or b < 0 or a >= n or b >= n for (a, b, *_) in graph.Edges):
            return Error("InvalidArgument", "Edge indices out of bounds")
        if any((len(e) > 2 and (e[2] < 0.0 or math.isnan(e[2]) or math.isinf(e[2]))) for e in graph.Edges):
            return Error("InvalidArgument", "Edge weights must be finite and non-negative")
        return Ok(None)

    @staticmethod
    def _validate(temperatures, graphs, iterations):
        if not Wrappers.available():
            return Error("DeviceNotAvailable", "GPU device not available")
        if any((t <= 0.0 or math.isnan(t) or math.isinf(t)) for t in temperatures):
            return Error("InvalidArgument", "Temperatures must be positive and finite")
        if iterations <= 0:
            return Error("InvalidArgument", f"Iterations must be positive, got {iterations}")
        if len(graphs) == 0:
            return Error("InvalidArgument", "No graphs provided")
        for g in graphs:
            r = GPUTempering.validateGraph(g)
            if r[0] == "error":
                return r
        return None

    @staticmethod
    def optimize(temperatures: Sequence[float], graphs: Sequence[SpatialGraph], iterations: int):
        """reference GPU.Wrappers.fs:271-344: flatten chain-major, pin, call the native step once per
        iteration, unflatten.  The topology of graphs[0] is shared by every chain (:307-308)."""
        bad = GPUTempering._validate(temperatures, graphs, iterations)
        if bad is not None:
            return bad
        try:
            lib = nt.load()
            node_count = len(graphs[0].Nodes)
            edge_count = len(graphs[0].Edges)
            T = len(temperatures)
            total = node_count * 3 * T
            if total > (2 ** 31 - 1) // 8:
                return Error("BufferOverflow", "Position buffer would exceed safe size")
            positions = np.zeros(total, np.float64)
            for i, g in enumerate(graphs):
                positions[i * node_count * 3:(i + 1) * node_count * 3] = Marshaling.flattenVec3(g.Positions)
            edges = Marshaling.flattenEdges(graphs[0].Edges)
            costs = np.zeros(T, np.float64)
            temps = np.asarray(temperatures, np.float64).copy()
            for _ in range(iterations):
                lib.parallel_tempering_step(nt.ptr(positions), nt.ptr(edges) if edge_count else None, nt.ptr(costs),
                                            nt.ptr(temps), node_count, edge_count, T)
                if lib.pt_last_error() != 0:
                    raise nt.PtError(lib.pt_last_error(), (lib.pt_last_error_string() or b"").decode())
            out = []
            for i, g in enumerate(graphs):
                p = Marshaling.unflattenVec3(positions[i * node_count * 3:(i + 1) * node_count * 3])
                out.append(SpatialGraph(g.Nodes, p, list(g.Edges)))
            GPUTempering.last_costs = costs
            return Ok(out)
        except Exception as ex:         # reference :343-344
            return Error("NativeException", str(ex))

    @staticmethod
    def optimize_batched(temperatures, graphs, iterations, *, seed="libgpu_compute/parallel_tempering_step"):
        """The same job in ONE native call through the batched engine: state stays in HBM for all
        `iterations` (the reference re-copies everything over PCIe every iteration)."""
        bad = GPUTempering._validate(temperatures, graphs, iterations)
        if bad is not None:
            return bad
        try:
            n, e, T = len(graphs[0].Nodes), len(graphs[0].Edges), len(temperatures)
            pos = np.stack([np.asarray(g.Positions, np.float32).reshape(n, 3) for g in graphs])[None]
            ed = np.tile(Marshaling.flattenEdges(graphs[0].Edges).reshape(1, 1, e, 2), (1, T, 1, 1)) if e else None
            with GraphEngine(n, e, T, 1, mutation_mode=nt.MUT_PERTURB_ONLY, seed=seed) as eng:
                eng.upload(pos, ed, np.full((1, T), e, np.int32), np.asarray(temperatures, np.float64)[None])
                eng.run(iterations)
                res = eng.download(edges=False)
            GPUTempering.last_costs = res["cost"][0]
            return Ok([SpatialGraph(g.Nodes, res["pos"][0, i], list(g.Edges)) for i, g in enumerate(graphs)])
        except Exception as ex:
            return Error("NativeException", str(ex))


class GPUCollision:
    @staticmethod
    def detect(minDistance: float, graph: SpatialGraph):
        """reference GPU.Wrappers.fs:381-416 (validation) and :349-379 (the native call): the pairs come
        back as (i, j) tuples.  Without a device the reference falls back to its CPU spatial hash; this
        library has no CPU path, so that case is an error here."""
        if minDistance <= 0.0 or math.isnan(minDistance) or math.isinf(minDistance):
            return Error("InvalidArgument", f"minDistance must be positive and finite, got {minDistance}")
        if len(graph.Positions) != len(graph.Nodes):
            return Error("InvalidArgument", f"Position count {len(graph.Positions)} != Node count {len(graph.Nodes)}")
        if not Wrappers.available():
            return Error("DeviceNotAvailable", "GPU device not available")
        try:
            lib = nt.load()
            n = len(graph.Nodes)
            max_collisions = n * n // 2
            if max_collisions > (2 ** 31 - 1) // 2:
                return Error("BufferOverflow", f"Too many potential collisions: {max_collisions}")
            positions = Marshaling.flattenVec3(graph.Positions)
            pairs = np.zeros(max(max_collisions, 1) * 2, np.int32)
            count = np.zeros(1, np.int32)
            lib.collision_detection(nt.ptr(positions), nt.ptr(pairs), n, float(minDistance), nt.ptr(count))
            if lib.pt_last_error() != 0:
                raise nt.PtError(lib.pt_last_error(), (lib.pt_last_error_string() or b"").decode())
            if count[0] < 0 or count[0] > max_collisions:
                return Error("BufferOverflow", f"Invalid collision count: {count[0]}")
            return Ok([(int(a), int(b)) for a, b in pairs[:2 * count[0]].reshape(-1, 2)])
        except Exception as ex:
            return Error("NativeException", str(ex))


@dataclass
class DispatchThresholds:
    NodeCount: int = 200          # reference GPU.Wrappers.fs:676-680
    EdgeCount: int = 500
    Iterations: int = 1000


@dataclass
class Chain:
    """reference src/ParallelTempering.Core.fs:22-27"""
    State: SpatialGraph
    Cost: float
    Temperature: float
    AcceptanceRate: float = 0.5


class Dispatch:
    _thresholds = DispatchThresholds()

    @classmethod
    def getThresholds(cls) -> DispatchThresholds:
        return cls._thresholds

    @classmethod
    def updateThresholds(cls, t: DispatchThresholds) -> None:
        if t.NodeCount > 0 and t.EdgeCount > 0 and t.Iterations > 0:            # reference :684-686
            cls._thresholds = t

    @classmethod
    def useGPU(cls, node_count: int, edge_count: int, iterations: int) -> bool:
        t = cls._thresholds                                                     # reference :689-693
        return (Wrappers.available() and node_count >= t.NodeCount and edge_count >= t.EdgeCount
                and iterations >= t.Iterations)

    @classmethod
    def optimize(cls, temperatures, graphs, mutate, cost: Callable[[SpatialGraph], float], iterations: int,
                 cpu_run: Optional[Callable] = None) -> List[Chain]:
        """reference GPU.Wrappers.fs:696-724.  `cpu_run(temperatures, graphs, mutate, cost, iterations, 10)`
        is the host's own ParallelTempering.run; this library has no CPU path of its own."""
        n, e = len(graphs[0].Nodes), len(graphs[0].Edges)
        if cls.useGPU(n, e, iterations):
            r = GPUTempering.optimize(temperatures, graphs, iterations)
            if r[0] == "ok":
                return [Chain(g, cost(g), t, 0.5) for t, g in zip(temperatures, r[1])]
        if cpu_run is None:
            raise RuntimeError("GPU path not taken and no CPU ParallelTempering.run supplied")
        return cpu_run(temperatures, graphs, mutate, cost, iterations, 10)

    @classmethod
    def optimizeUntilConvergence(cls, temperatures, graphs, maxIterations: int, strictness: float,
                                 onProgress: Optional[Callable] = None, *, node_kind=None, node_value=None,
                                 edge_capacity: int = 0, seed="libgpu_compute/optimize_until_convergence"):
        """reference GPU.Wrappers.fs:728-753 over Convergence.runUntilConvergence
        (ParallelTempering.Core.fs:496-550), with the whole loop on the device: the engine's adaptive
        mutation and procedural cost take the place of the host's `mutate` and `cost` closures, the
        swap frequency and the check frequency are the reference's (:740-741).  `onProgress(step,
        chains, metrics)` is called once, for the check that ended the run (the reference calls it at
        every check).  Returns (chains, metrics) like the reference."""
        n, T = len(graphs[0].Nodes), len(temperatures)
        emax = max(len(g.Edges) for g in graphs)
        ecap = edge_capacity or max(2 * emax, emax + 64)
        check_freq = max(50, maxIterations // 20)
        pos = np.stack([np.asarray(g.Positions, np.float32).reshape(n, 3) for g in graphs])[None]
        ed = np.zeros((1, T, ecap, 2), np.int32)
        ec = np.zeros((1, T), np.int32)
        for i, g in enumerate(graphs):
            if g.Edges:
                ed[0, i, :len(g.Edges)] = Marshaling.flattenEdges(g.Edges).reshape(-1, 2)
            ec[0, i] = len(g.Edges)
        with GraphEngine(n, ecap, T, 1, seed=seed) as eng:
            eng.upload(pos, ed, ec, np.asarray(temperatures, np.float64)[None], node_kind, node_value)
            steps, metrics = eng.run_until_converged(maxIterations, check_freq, strictness)
            res = eng.download()
            st = eng.stats()
        if st["status_bits"]:
            raise RuntimeError(f"device reported status bits {st['status_bits']:#x}")
        chains = []
        for i, g in enumerate(graphs):
            k = int(res["ecount"][0, i])
            edges = [(int(a), int(b), 1.0) for a, b in res["edges"][0, i, :k]]
            chains.append(Chain(SpatialGraph(g.Nodes, res["pos"][0, i], edges), float(res["cost"][0, i]),
                                float(temperatures[i]), 0.5))
        if onProgress is not None:
            onProgress(steps, chains, metrics[0])
        return chains, metrics[0]

"""Host-side dispatch layer, CPU only: the cases of the reference's tests/GPUIntegrationTests.fs
that need no device (threshold clamps :27-47, marshaling round trips :75-112, argument
validation :155-226, dispatch thresholds :229-258)."""
import math

import numpy as np

from procgen_amd.wrappers import (Dispatch, DispatchThresholds, GPUTempering, Marshaling, SpatialGraph, Wrappers)


def _graph(n=3, edges=((0, 1, 1.0), (1, 2, 1.0))):
    pos = np.zeros((n, 3), np.float32); pos[:, 0] = np.arange(n)
    return SpatialGraph(list(range(1, n + 1)), pos, list(edges))


def test_gpu_threshold_is_clamped():
    assert Wrappers.getGPUThreshold() > 0
    Wrappers.updateGPUThreshold(500)
    assert 1000 <= Wrappers.getGPUThreshold() <= 1000000
    Wrappers.updateGPUThreshold(100)
    assert Wrappers.getGPUThreshold() >= 1000
    Wrappers.updateGPUThreshold(10000000)
    assert Wrappers.getGPUThreshold() <= 1000000
    Wrappers.updateGPUThreshold(10000)


def test_availability_probe_returns_a_bool():
    assert Wrappers.available() in (True, False)
    assert Wrappers.shouldUseGPU(100) is False


def test_marshaling_roundtrip():
    v = np.array([[1, 2, 3], [4, 5, 6]], np.float32)
    flat = Marshaling.flattenVec3(v)
    assert flat.dtype == np.float64 and flat.size == 6
    assert np.array_equal(Marshaling.unflattenVec3(flat), v)
    assert Marshaling.flattenEdges([(0, 1, 1.0), (1, 2, 2.0), (2, 3, 3.0)]).tolist() == [0, 1, 1, 2, 2, 3]
    assert Marshaling.toVector3Array([1.0, 2, 3, 4, 5, 6])[0] == "ok"
    r = Marshaling.toVector3Array([1.0, 2, 3, 4])
    assert r[0] == "error" and r[1].kind == "InvalidArgument"


def test_validate_graph():
    assert GPUTempering.validateGraph(_graph())[0] == "ok"
    g = _graph(); g.Positions = g.Positions[:2]
    assert GPUTempering.validateGraph(g)[1].kind == "InvalidArgument"           # mismatched counts
    assert GPUTempering.validateGraph(_graph(edges=[(0, 5, 1.0)]))[1].kind == "InvalidArgument"
    assert GPUTempering.validateGraph(_graph(edges=[(0, 1, math.nan)]))[1].kind == "InvalidArgument"
    assert GPUTempering.validateGraph(_graph(edges=[(0, 1, -1.0)]))[1].kind == "InvalidArgument"


def test_dispatch_thresholds():
    t = Dispatch.getThresholds()
    assert t.NodeCount > 0 and t.EdgeCount > 0 and t.Iterations > 0
    Dispatch.updateThresholds(DispatchThresholds(-1, 100, 100))
    assert Dispatch.getThresholds().NodeCount > 0
    assert Dispatch.useGPU(t.NodeCount - 1, t.EdgeCount - 1, t.Iterations - 1) is False


def test_dispatch_falls_back_to_the_hosts_cpu_run():
    # reference GPUIntegrationTests.fs:260-275: a small problem goes to ParallelTempering.run
    calls = []

    def cpu_run(temps, graphs, mutate, cost, iters, swap_freq):
        calls.append((len(temps), iters, swap_freq))
        return ["chain"] * len(temps)

    out = Dispatch.optimize([1.0, 0.5], [_graph(), _graph()], lambda g, t: g, lambda g: 0.0, 10, cpu_run=cpu_run)
    assert len(out) == 2 and calls == [(2, 10, 10)]

"""The drop-in entry point the F# host binds (parallel_tempering_step), driven the way
GPUTempering.optimize drives it (reference src/GPU.Wrappers.fs:271-344), against the oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
import parity_util as pu
from procgen_amd import native as nt
from procgen_amd import workloads as wlm
from procgen_amd.wrappers import Dispatch, DispatchThresholds, GPUTempering, SpatialGraph, Wrappers

pytestmark = pytest.mark.gpu
SEED = "libgpu_compute/parallel_tempering_step"


def _job(n, T, name="legacy"):
    wl = wlm.graph_workload(name, n, T, 1)
    e = int(wl["ecount"][0, 0])
    wl["value"][:] = 0.0                    # node values do not cross the legacy ABI
    edges = [(int(a), int(b), 1.0) for a, b in wl["edges"][0, 0, :e]]
    graphs = [SpatialGraph(list(range(n)), wl["pos"][0, t].copy(), list(edges)) for t in range(T)]
    wl["ecap"] = e
    wl["edges"] = np.ascontiguousarray(wl["edges"][:, :, :e])
    return wl, graphs


def _oracle(wl, iters, f):
    return pu.oracle_run(wl, iters, f, seed=SEED, mutation_mode=ob.MUT_PERTURB_ONLY, trace=False)[0]


def test_probe_then_available(gpu_lib):
    gpu_lib.parallel_tempering_step(None, None, None, None, 0, 0, 0)
    assert gpu_lib.pt_last_error() == 0
    assert Wrappers.available() is True


@pytest.mark.parametrize("n,T,iters", [(60, 4, 40), (250, 5, 70)])
def test_step_per_iteration_matches_oracle(gpu_lib, n, T, iters):
    gpu_lib.parallel_tempering_reset()
    wl, graphs = _job(n, T)
    temps = wl["temperatures"][0]
    r = GPUTempering.optimize(temps, graphs, iters)
    assert r[0] == "ok", r[1]
    f = nt.swap_frequency(n)
    o = _oracle(wl, iters, f)
    got = np.stack([g.Positions for g in r[1]])
    assert np.array_equal(got, o["pos"])
    assert np.array_equal(GPUTempering.last_costs, o["cost"])
    # one native call for the whole job gives the same bits
    rb = GPUTempering.optimize_batched(temps, graphs, iters)
    assert rb[0] == "ok"
    assert np.array_equal(np.stack([g.Positions for g in rb[1]]), o["pos"])
    assert np.array_equal(GPUTempering.last_costs, o["cost"])


def test_f32_twin_and_state_reset(gpu_lib):
    n, T, iters = 40, 3, 25
    wl, graphs = _job(n, T, "legacy32")
    e = int(wl["ecount"][0, 0])
    pos = wl["pos"][0].astype(np.float32).reshape(-1).copy()
    edges = wl["edges"][0, 0].reshape(-1).astype(np.int32).copy()
    costs = np.zeros(T, np.float32)
    temps = wl["temperatures"][0].astype(np.float32)
    gpu_lib.parallel_tempering_reset()
    for _ in range(iters):
        gpu_lib.parallel_tempering_step_f32(nt.ptr(pos), nt.ptr(edges), nt.ptr(costs), nt.ptr(temps), n, e, T)
        assert gpu_lib.pt_last_error() == 0
    wl["temperatures"] = temps.astype(np.float64)[None]      # the ladder as the f32 caller sees it
    o = _oracle(wl, iters, nt.swap_frequency(n))
    assert np.array_equal(pos.reshape(T, n, 3), o["pos"])
    assert np.array_equal(costs, o["cost"].astype(np.float32))
    # a different shape restarts the call counter and the streams
    gpu_lib.parallel_tempering_reset()
    pos2 = wl["pos"][0].astype(np.float64).reshape(-1).copy()
    c2 = np.zeros(T, np.float64); t2 = temps.astype(np.float64)
    gpu_lib.parallel_tempering_step(nt.ptr(pos2), nt.ptr(edges), nt.ptr(c2), nt.ptr(t2), n, e, T)
    o1 = _oracle(wl, 1, nt.swap_frequency(n))
    assert np.array_equal(pos2.reshape(T, n, 3).astype(np.float32), o1["pos"])


def test_bad_arguments_leave_outputs_untouched(gpu_lib):
    pos = np.ones(9, np.float64); costs = np.full(1, -7.0); temps = np.array([1.0])
    edges = np.array([0, 5], np.int32)                                   # index out of bounds
    gpu_lib.parallel_tempering_reset()
    gpu_lib.parallel_tempering_step(nt.ptr(pos), nt.ptr(edges), nt.ptr(costs), nt.ptr(temps), 3, 1, 1)
    assert gpu_lib.pt_last_error() == 1 and costs[0] == -7.0 and np.all(pos == 1.0)
    gpu_lib.parallel_tempering_step(nt.ptr(pos), None, nt.ptr(costs), nt.ptr(temps), -3, 0, 1)
    assert gpu_lib.pt_last_error() == 1
    temps[0] = -1.0
    gpu_lib.parallel_tempering_step(nt.ptr(pos), None, nt.ptr(costs), nt.ptr(temps), 3, 0, 1)
    assert gpu_lib.pt_last_error() == 1 and costs[0] == -7.0


def test_wrapper_validation_on_device(gpu_lib):
    # reference GPUIntegrationTests.fs:155-184
    _, graphs = _job(10, 1, "val")
    assert GPUTempering.optimize([-1.0], graphs, 100)[1].kind == "InvalidArgument"
    assert GPUTempering.optimize([float("nan")], graphs, 100)[1].kind == "InvalidArgument"
    assert GPUTempering.optimize([1.0], graphs, -1)[1].kind == "InvalidArgument"
    assert GPUTempering.optimize([1.0], [], 10)[1].kind == "InvalidArgument"


def test_dispatch_takes_the_gpu_path_above_thresholds(gpu_lib):
    gpu_lib.parallel_tempering_reset()
    n, T = 220, 3
    wl, graphs = _job(n, T, "dispatch")
    assert len(graphs[0].Edges) < Dispatch.getThresholds().EdgeCount
    Dispatch.updateThresholds(DispatchThresholds(200, 300, 20))
    try:
        chains = Dispatch.optimize(wl["temperatures"][0], graphs, None, lambda g: 1.5, 20)
    finally:
        Dispatch.updateThresholds(DispatchThresholds())
    assert len(chains) == T and all(c.Cost == 1.5 and c.AcceptanceRate == 0.5 for c in chains)
    o = _oracle(wl, 20, nt.swap_frequency(n))
    assert np.array_equal(np.stack([c.State.Positions for c in chains]), o["pos"])


def test_resident_chains_and_a_caller_that_edits_its_positions(gpu_lib):
    """The entry keeps the chains on the device between calls and checks that the caller's array still holds
    them; a caller that edits positions between two calls gets exactly what a fresh upload of the edited
    array gives."""
    n, T = 80, 4
    f = nt.swap_frequency(n)
    assert f == 17
    wl, _ = _job(n, T, "edit")
    e = int(wl["ecount"][0, 0])
    pos = wl["pos"][0].astype(np.float64).reshape(-1).copy()
    edges = wl["edges"][0, 0].reshape(-1).astype(np.int32).copy()
    costs = np.zeros(T, np.float64)
    temps = wl["temperatures"][0].astype(np.float64).copy()

    def call(k):
        for _ in range(k):
            gpu_lib.parallel_tempering_step(nt.ptr(pos), nt.ptr(edges), nt.ptr(costs), nt.ptr(temps), n, e, T)
            assert gpu_lib.pt_last_error() == 0, gpu_lib.pt_last_error_string()

    gpu_lib.parallel_tempering_reset()
    call(12)                                            # calls 0..11 (a swap pass after call 0)
    cfg = ob.default_config(ob.SUM_CANONICAL, ob.ALGO_OPTIMISED, ob.MUT_PERTURB_ONLY)
    key = int(ob.lib().og_seed_key(SEED.encode()))
    o1 = ob.run(wl["pos"][0], wl["edges"][0], wl["ecount"][0], wl["kind"][0], wl["value"][0], temps, 12, f, cfg, key=key,
                trace=False)
    assert np.array_equal(pos.reshape(T, n, 3).astype(np.float32), o1["pos"]) and np.array_equal(costs, o1["cost"])
    # the caller moves one node of chain 2 and stretches chain 0
    p3 = pos.reshape(T, n, 3)
    p3[2, 5] += 0.25
    p3[0] *= 1.01
    edited = p3.astype(np.float32)
    call(4)                                             # calls 12..15: no swap pass among them
    o2 = ob.run(edited, o1["edges"], o1["ecount"], wl["kind"][0], wl["value"][0], temps, 4, f, cfg, key=key, trace=False,
                first_step=12)
    assert np.array_equal(pos.reshape(T, n, 3).astype(np.float32), o2["pos"]) and np.array_equal(costs, o2["cost"])
    # a caller that changes the ladder or the topology is followed as well
    temps[:] = temps * 1.5
    edges2 = edges.copy()
    edges2[0:2] = [3, 9] if not (edges[0] == 3 and edges[1] == 9) else [4, 9]
    before = pos.reshape(T, n, 3).astype(np.float32).copy()
    gpu_lib.parallel_tempering_step(nt.ptr(pos), nt.ptr(edges2), nt.ptr(costs), nt.ptr(temps), n, e, T)
    assert gpu_lib.pt_last_error() == 0
    ed = np.tile(edges2.reshape(1, e, 2), (T, 1, 1))
    o3 = ob.run(before, ed, o1["ecount"], wl["kind"][0], wl["value"][0], temps, 1, f, cfg, key=key, trace=False, first_step=16)
    assert np.array_equal(pos.reshape(T, n, 3).astype(np.float32), o3["pos"]) and np.array_equal(costs, o3["cost"])
    gpu_lib.parallel_tempering_reset()


_FREED_ARRAY_JOB = r"""
import gc, os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import oracle_binding as ob
from procgen_amd import native as nt
L = nt.load()
pin = os.environ.get("PROCGEN_PT_PIN_CALLER_ARRAY") == "1"
n, T, E = 3000, 8, 4000                       # 576 KB of positions: malloc takes it from mmap and gives it back on free
rng = np.random.default_rng(3)
edges = np.ascontiguousarray(np.stack([np.arange(E) % n, (np.arange(E) * 7 + 1) % n], 1).astype(np.int32))
temps = np.ascontiguousarray(nt.geometric_ladder(0.5, 5.0, T), np.float64)
def job(calls):
    pos = np.ascontiguousarray(np.tile(rng.uniform(0, 9, (1, n, 3)), (T, 1, 1)))
    costs = np.zeros(T)
    for _ in range(calls):
        L.parallel_tempering_step(nt.ptr(pos), nt.ptr(edges), nt.ptr(costs), nt.ptr(temps), n, E, T)
        assert L.pt_last_error() == 0, L.pt_last_error_string()
    assert np.isfinite(costs).all() and (costs > 0).all()
    return pos
pos = job(3)                                  # the same array three times in a row
if pin:
    L.parallel_tempering_reset()              # the documented duty of a host that opted in
del pos; gc.collect()
# host allocations of the same size class follow: pageable buffers that may land where the array was
for k in range(3):
    pts = np.ascontiguousarray(rng.uniform(0, 6, (n, 3)).astype(np.float32).astype(np.float64))
    pairs = np.zeros((n * n // 2, 2), np.int32)
    count = np.zeros(1, np.int32)
    L.collision_detection(nt.ptr(pts), nt.ptr(pairs), n, 0.5, nt.ptr(count))
    assert L.pt_last_error() == 0, L.pt_last_error_string()
    want = ob.collision_pairs(pts.astype(np.float32), 0.5)
    assert count[0] == len(want) > 0 and np.array_equal(pairs[:count[0]], want), k
    del pairs
pos = job(2)                                  # a new job, new array
print("FREED_ARRAY_OK")
"""


@pytest.mark.parametrize("pin", ["default", "opt-in pinning"])
def test_callers_array_freed_between_jobs(gpu_lib, pin):
    """The host only pins its positions array for the job (reference src/GPU.Wrappers.fs:312-325) and may free it
    afterwards: nothing of it is retained by default, so later calls that stage through pageable host buffers
    (collision_detection) and a new job run cleanly; with PROCGEN_PT_PIN_CALLER_ARRAY=1 the same holds once the
    host has called parallel_tempering_reset before freeing."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("PROCGEN_PT_PIN_CALLER_ARRAY", None)
    if pin != "default":
        env["PROCGEN_PT_PIN_CALLER_ARRAY"] = "1"
    out = subprocess.run([sys.executable, "-c", _FREED_ARRAY_JOB.format(root=root)], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and "FREED_ARRAY_OK" in out.stdout, out.stderr[-3000:]

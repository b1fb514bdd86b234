"""Failures a kernel can only flag on the device must come back as return codes from the next call
that waits for the stream (sync, download, energies, convergence check, upload), and the drop-in
entry must leave its outputs untouched."""
import numpy as np
import pytest

from procgen_amd import native as nt
from procgen_amd import workloads as wlm
from procgen_amd.engine import GraphEngine

pytestmark = pytest.mark.gpu


def test_injected_draws_shorter_than_the_run(gpu_lib):
    n, T, f = 40, 3, 5
    wl = wlm.graph_workload("short-draws", n, T, 1)
    rng = np.random.default_rng(1)
    with GraphEngine(n, wl["ecap"], T, 1, swap_frequency=f, rng_mode=nt.RNG_INJECTED) as eng:
        eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
        eng.set_draws(rng.random((1, T, 10, nt.DRAWS_PER_STEP)), rng.random((1, 4, T - 1)))
        eng.run(10)
        eng.sync()                                      # exactly as many draws as steps: fine
        eng.run(6)
        with pytest.raises(nt.PtError) as exc:
            eng.sync()
        assert exc.value.code == 6
        with pytest.raises(nt.PtError) as exc:
            eng.download()
        assert exc.value.code == 6
        assert eng.stats()["status_bits"] & (1 << 6)
        # a fresh upload clears the condition
        eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
        eng.sync()


@pytest.mark.parametrize("flags", [0, nt.FLAG_FULL_EVALUATION, nt.FLAG_STREAMING_INCREMENTAL])
def test_cost_term_out_of_range_is_an_error(gpu_lib, flags):
    n, T = 50, 2
    wl = wlm.graph_workload("range", n, T, 1)
    wl["pos"][0, :, 3] = 1.0e19                          # an edge of length 1e19: (d - 2)^2 is far beyond 2^62
    with GraphEngine(n, wl["ecap"], T, 1, seed="range", flags=flags) as eng:
        with pytest.raises(nt.PtError) as exc:
            eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
        assert exc.value.code == 4


def test_legacy_entry_leaves_outputs_untouched_on_a_device_error(gpu_lib):
    lib = nt.load()
    n, T = 60, 3
    wl = wlm.graph_workload("legacy-range", n, T, 1)
    pos = np.ascontiguousarray(wl["pos"][0], np.float64).copy()
    pos[1, 5] = 1.0e19
    before = pos.copy()
    ec = int(wl["ecount"][0, 0])
    edges = np.ascontiguousarray(wl["edges"][0, 0, :ec], np.int32)
    costs = np.full(T, -3.0)
    temps = np.ascontiguousarray(wl["temperatures"][0], np.float64)
    lib.parallel_tempering_step(nt.ptr(pos), nt.ptr(edges), nt.ptr(costs), nt.ptr(temps), n, ec, T)
    assert lib.pt_last_error() == 4, lib.pt_last_error_string()
    assert np.array_equal(pos, before) and (costs == -3.0).all()
    lib.parallel_tempering_reset()

"""Convergence monitoring on the device (history rings written by the step and swap kernels, the
check kernel, the run-until-converged loop and the legacy extern) against the oracle's restatement
of Convergence.updateHistory / checkConvergence (reference src/ParallelTempering.Core.fs:342-492).

Bar: every statistic bit-exact (both sides sum in the reference's order).  The history the oracle
is fed is rebuilt from the engine's own step trace and swap trace, which the graph parity tests pin
against the oracle's PT loop.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from procgen_amd import native as nt
from procgen_amd import workloads as wlm
from procgen_amd.engine import GraphEngine

pytestmark = pytest.mark.gpu

KEYS = ("gelman_rubin", "effective_sample_size", "cost_stabilization", "swap_acceptance_rate", "has_converged")


def same(got, want, where=""):
    for k in KEYS:
        assert got[k] == want[k], (where, k, got[k], want[k])


def history_rows(trace, swaps, g, T, f, first_step, last_step):
    """Chain costs per slot after every step (after the swap pass on swap steps) and the outcomes."""
    for s in range(first_step, last_step):
        row = np.array(trace["cost_after"][s, g * T:(g + 1) * T], np.float64)
        outs = ()
        if s % f == 0 and T > 1:
            outs = swaps[s // f, g]
            for i in range(T - 1):
                if outs[i]:
                    row[i], row[i + 1] = row[i + 1], row[i]
        yield row, outs


CONFIGS = [pytest.param(0, 0, id="incremental"), pytest.param(nt.FLAG_FULL_EVALUATION, 0, id="full-lds"),
           pytest.param(nt.FLAG_GLOBAL_MEMORY_PLACEMENT, 0, id="full-global"),
           pytest.param(nt.FLAG_FULL_EVALUATION, 2, id="cluster2"),
           pytest.param(nt.FLAG_STREAMING_INCREMENTAL, 1, id="streaming"),
           pytest.param(nt.FLAG_STREAMING_INCREMENTAL, 3, id="streaming-cluster3")]


@pytest.mark.parametrize("flags,cluster", CONFIGS)
def test_history_and_check_match_the_oracle(gpu_lib, flags, cluster):
    T, G, f, window, total = 4, 2, 7, 30, 110
    wl = wlm.graph_workload("conv", 60, T, G, edge_headroom=3.0)
    with GraphEngine(wl["n"], wl["ecap"], T, G, swap_frequency=f, seed="conv-test", flags=flags, cluster_size=cluster) as eng:
        eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
        eng.enable_trace(total, total // f + 1)
        eng.enable_convergence(window)
        hist = [ob.ConvHistory(T, window) for _ in range(G)]
        done = 0
        for upto in (5, 12, 29, 30, 31, 64, 110):
            eng.run(upto - done)
            got = eng.check_convergence(0.25, iteration=upto)
            tr, sw = eng.read_trace(total, total // f + 1)
            for g in range(G):
                for row, outs in history_rows(tr, sw, g, T, f, done, upto):
                    hist[g].update(row, outs)
                want = hist[g].check(upto, 0.25)
                same(got[g], want, f"after {upto} steps, ladder {g}")
                assert got[g]["iteration"] == upto
            done = upto
        assert eng.stats()["status_bits"] == 0
        # the statistics are not degenerate
        assert got[0]["gelman_rubin"] not in (None, 1.0) and got[0]["effective_sample_size"] is not None


def test_history_starts_where_it_is_enabled(gpu_lib):
    T, f, window = 3, 5, 40
    wl = wlm.graph_workload("conv-late", 40, T, 1, edge_headroom=3.0)
    with GraphEngine(wl["n"], wl["ecap"], T, 1, swap_frequency=f, seed="conv-late") as eng:
        eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
        eng.enable_trace(100, 21)
        eng.run(33)
        eng.enable_convergence(window)
        eng.run(25)
        got = eng.check_convergence(0.0)[0]
        tr, sw = eng.read_trace(100, 21)
        h = ob.ConvHistory(T, window)
        for row, outs in history_rows(tr, sw, 0, T, f, 33, 58):
            h.update(row, outs)
        same(got, h.check(0, 0.0))
        # enabling again empties the history
        eng.enable_convergence(window)
        got = eng.check_convergence(0.0)[0]
        assert got["gelman_rubin"] is None and got["effective_sample_size"] is None
        assert got["cost_stabilization"] == 1.0 and got["swap_acceptance_rate"] == 0.0


def test_run_until_converged_follows_the_reference_schedule(gpu_lib):
    """runUntilConvergence (reference src/ParallelTempering.Core.fs:496-550): window max(200, max/10),
    checks after step s when s % checkFreq == 0 and s > window/2, one final check at max."""
    T, G, f, max_it, cf = 4, 2, 10, 260, 50
    wl = wlm.graph_workload("conv-run", 50, T, G, edge_headroom=3.0)

    def make():
        e = GraphEngine(wl["n"], wl["ecap"], T, G, swap_frequency=f, seed="conv-run")
        e.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
        e.enable_trace(max_it, max_it // f + 1)
        return e

    with make() as eng:
        steps, got = eng.run_until_converged(max_it, cf, 0.0)
        tr, sw = eng.read_trace(max_it, max_it // f + 1)
        final_cost = eng.download(positions=False, edges=False)["cost"]
    # the host loop of the reference, driven by the recorded costs
    window = max(200, max_it // 10)
    hist = [ob.ConvHistory(T, window) for _ in range(G)]
    rows = [list(history_rows(tr, sw, g, T, f, 0, max_it)) for g in range(G)]
    want, want_steps = None, max_it
    for s in range(max_it):
        for g in range(G):
            hist[g].update(*rows[g][s])
        if s % cf == 0 and s > window // 2:
            want = [hist[g].check(s, 0.0) for g in range(G)]
            if all(w["has_converged"] for w in want):
                want_steps = s + 1
                break
    else:
        want = [hist[g].check(max_it, 0.0) for g in range(G)]
    assert steps == want_steps
    for g in range(G):
        same(got[g], want[g], f"ladder {g}")
        assert got[g]["iteration"] == want[g]["iteration"]
    # the same job through run() ends in the same state
    with make() as eng:
        eng.run(steps)
        assert np.array_equal(eng.download(positions=False, edges=False)["cost"], final_cost)


def test_run_until_converged_stops_early(gpu_lib):
    """With every weight zero the cost is 0 for good: R-hat = 1 (W = 0), ESS = n, CV = 0.  Equal costs
    swap with probability 1 (reference src/ParallelTempering.Core.fs:112-119), so an injected swap
    draw of exactly 1.0 is the only one that fails: the pattern 0.5, 1, 1 swaps one pair in three."""
    T, f, cf, max_it = 4, 1, 10, 1000
    n, draws = 3, 200
    pos = np.zeros((1, T, n, 3), np.float32)
    pos[..., 0] = np.arange(n) * 10.0
    rng = np.random.default_rng(0)
    with GraphEngine(n, 8, T, 1, swap_frequency=f, rng_mode=nt.RNG_INJECTED, mutation_mode=nt.MUT_PERTURB_ONLY,
                     weights=[0, 0, 0, 0, 0, 0]) as eng:
        eng.upload(pos, None, None, np.array([[1.0, 2.0, 3.0, 4.0]]))
        eng.set_draws(rng.random((1, T, draws, 6)), np.tile([0.5, 1.0, 1.0], (1, draws, 1)))
        steps, got = eng.run_until_converged(max_it, cf, 1.0)
        assert eng.stats()["status_bits"] == 0
    m = got[0]
    # 111 passes append (pair 2, pair 1, pair 0) = (0, 0, 1); the last 200 outcomes hold 67 swaps
    assert m["swap_acceptance_rate"] == 67.0 / 200.0
    assert m["gelman_rubin"] == 1.0 and m["cost_stabilization"] == 0.0
    assert m["has_converged"]
    # window = max(200, 100): the first check is after step 110
    assert steps == 111 and m["iteration"] == 110 and m["effective_sample_size"] == 111.0


def _legacy(cost, swaps, strictness, iteration):
    L = nt.load()
    m = nt.LegacyConvergenceMetrics()
    c = np.ascontiguousarray(cost, np.float64)
    s = np.ascontiguousarray(swaps, np.int32)
    L.check_convergence_async(c.ctypes.data if c.size else None, s.ctypes.data if s.size else None,
                              c.shape[0] if c.ndim == 2 else 0, c.shape[1] if c.ndim == 2 else 0, s.size,
                              strictness, iteration, C.byref(m), None)
    assert L.pt_last_error() == 0
    return m


def test_legacy_extern(gpu_lib):
    rng = np.random.default_rng(2)
    for chains, n in ((2, 10), (5, 64), (3, 200), (70, 33)):
        x = rng.normal(80.0, 2.0, (chains, n)) + rng.normal(0, 1, (chains, 1))
        sw = rng.integers(0, 2, 37) * rng.integers(1, 5, 37)          # any non-zero value is true
        m = _legacy(x, sw, 0.4, 12)
        h = ob.ConvHistory(chains, n)
        for s in range(n):
            h.update(x[:, s])
        # the swap history arrives as it is (oldest first)
        hs = ob.ConvHistory(1, 10 ** 6)
        for v in sw:
            hs.update([0.0], [1 if v else 0])
        want = h.check(12, 0.4)
        rate = hs.check(0, 0)["swap_acceptance_rate"]
        assert m.GelmanRubin == np.float32(want["gelman_rubin"])
        assert m.EffectiveSampleSize == np.float32(want["effective_sample_size"] if want["effective_sample_size"] is not None else 1.0)
        assert m.CostStabilization == np.float32(want["cost_stabilization"])
        assert m.AvgSwapAcceptance == np.float32(rate)
        assert m.Iteration == 12
    # a converged history: white noise around 100, a third of the swaps taken
    x = rng.normal(100.0, 1.0, (4, 200))
    m = _legacy(x, [1, 0, 0] * 20, 0.0, 5)
    assert m.HasConverged == 1 and m.GelmanRubin < 1.2 and m.EffectiveSampleSize > 50
    assert _legacy(x, [1, 1, 0] * 20, 0.0, 5).HasConverged == 0
    # too little history: the placeholders of the CUDA routine
    m = _legacy(np.zeros((1, 5)), [], 0.0, 3)
    assert (m.GelmanRubin, m.EffectiveSampleSize, m.HasConverged, m.Iteration) == (2.0, 1.0, 0, 3)
    m = _legacy(np.zeros((0, 0)), [], 0.0, 3)
    assert (m.GelmanRubin, m.EffectiveSampleSize, m.CostStabilization, m.AvgSwapAcceptance) == (2.0, 1.0, 1.0, 0.0)


def test_optimize_until_convergence_wrapper(gpu_lib):
    """Dispatch.optimizeUntilConvergence (reference src/GPU.Wrappers.fs:728-753): the chains handed
    back carry the cost of the state handed back."""
    from procgen_amd.wrappers import Dispatch, SpatialGraph
    n, T = 40, 4
    wl = wlm.graph_workload("conv-wrap", n, T, 1, edge_headroom=3.0)
    graphs = []
    for t in range(T):
        k = int(wl["ecount"][0, t])
        graphs.append(SpatialGraph(list(range(n)), wl["pos"][0, t], [(int(a), int(b), 1.0) for a, b in wl["edges"][0, t, :k]]))
    seen = []
    chains, metrics = Dispatch.optimizeUntilConvergence(wl["temperatures"][0], graphs, 400, 0.0,
                                                        lambda s, c, m: seen.append((s, len(c), m)))
    assert len(chains) == T and len(seen) == 1 and seen[0][1] == T
    assert metrics["iteration"] <= 400 and metrics["gelman_rubin"] is not None
    cfg = ob.default_config()
    for c, t in zip(chains, wl["temperatures"][0]):
        assert c.Temperature == t
        ed = np.array([(a, b) for a, b, _ in c.State.Edges], np.int32).reshape(-1, 2)
        want = ob.cost(c.State.Positions, ed, ob.default_kinds(n), np.zeros(n), cfg).total
        assert c.Cost == want


@pytest.mark.parametrize("T", [64, 65, 66, 129, 200])
def test_long_ladders_swap_pass_and_history(gpu_lib, T):
    """Ladders longer than a wave: the swap pass takes the pairs 64 at a time and carries the state that
    bubbles up from one batch into the next.  Injected swap draws force refusals on the first and the last
    pair of a batch and unbroken runs of swaps across the batch boundary; every decision, the final slot
    map, the costs and the convergence statistics equal the oracle's."""
    import parity_util as pu
    n, G, f, iters, window = 12, 2, 3, 45, 20
    wl = wlm.graph_workload("long-ladder", n, T, G, edge_headroom=6.0)
    wl["temperatures"][:] = nt.geometric_ladder(0.3, 9.0, T)
    rng = np.random.default_rng(T)
    passes = (iters + f - 1) // f
    chain_u = rng.random((G, T, iters, nt.DRAWS_PER_STEP))
    swap_u = rng.random((G, passes, T - 1))
    never, always = np.nextafter(1.0, 0.0), 0.0
    for p in range(passes):
        mode = p % 5
        if mode == 0:
            swap_u[:, p, :] = always                    # one state carried from the hottest slot to the coldest
        elif mode == 1:
            swap_u[:, p, :] = always
            swap_u[:, p, 0] = never                      # refusal on the first pair of the first batch
            if T > 64:
                swap_u[:, p, 63] = never                 # ... on the last pair of the first batch
        elif mode == 2 and T > 65:
            swap_u[:, p, :] = never
            swap_u[:, p, 60:min(70, T - 1)] = always     # a run of swaps across the batch boundary
        elif mode == 3 and T > 65:
            swap_u[:, p, 64] = never                     # refusal on the first pair of the second batch
    with GraphEngine(n, wl["ecap"], T, G, swap_frequency=f, rng_mode=nt.RNG_INJECTED) as eng:
        eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
        eng.set_draws(chain_u, swap_u)
        eng.enable_trace(iters, passes)
        eng.enable_convergence(window)
        eng.run(20)
        eng.run(iters - 20)
        got = eng.check_convergence(0.25, iteration=iters)
        gpu = eng.download()
        gpu["trace"], gpu["swap_trace"] = eng.read_trace(iters, passes)
        gpu["stats"], gpu["slots"], gpu["terms"] = eng.stats(), eng.replica_slots(), eng.cost_terms()
    orc = pu.oracle_run(wl, iters, f, chain_u=chain_u, swap_u=swap_u)
    pu.assert_same_run(wl, gpu, orc, iters)
    for g in range(G):
        assert np.array_equal(gpu["slots"][g], pu.slots_after(orc[g]["swap_trace"], T))
        h = ob.ConvHistory(T, window)
        for row, outs in history_rows(gpu["trace"], gpu["swap_trace"], g, T, f, 0, iters):
            h.update(row, outs)
        same(got[g], h.check(iters, 0.25), f"ladder {g}")
    assert gpu["swap_trace"].any() and not gpu["swap_trace"].all()

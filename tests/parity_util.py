"""Shared helpers of the GPU parity tests: run the same job through the HIP
engine (C ABI) and through the CPU oracle, and compare."""
import numpy as np

import oracle_binding as ob
from procgen_amd import native as nt
from procgen_amd.engine import GraphEngine

TRACE_FIELDS = ["kind", "a", "b", "flags", "reach0", "reach_origin", "delta", "u", "cost_after"]


def oracle_run(wl, iters, swap_freq, *, seed=None, key=None, mutation_mode=ob.MUT_ADAPTIVE, weights=None,
               chain_u=None, swap_u=None, algo=ob.ALGO_OPTIMISED, sum_mode=ob.SUM_CANONICAL, trace=True, threads=1):
    """Run every ladder of a workload through the oracle.  Returns a list per ladder."""
    if key is None and seed is not None:
        key = int(ob.lib().og_seed_key(seed.encode()))
    cfg = ob.default_config(sum_mode, algo, mutation_mode, weights)
    out = []
    T = wl["temps"]
    for g in range(wl["groups"]):
        kw = {}
        if chain_u is not None:
            kw = dict(chain_u=chain_u[g].reshape(T, -1), swap_u=swap_u[g].reshape(-1))
        out.append(ob.run(wl["pos"][g], wl["edges"][g], wl["ecount"][g], wl["kind"][g], wl["value"][g],
                          wl["temperatures"][g], iters, swap_freq, cfg, key=key, stream_base=g * T,
                          swap_stream_id=g, trace=trace, threads=threads, **kw))
    return out


def gpu_run(wl, iters, swap_freq, *, seed=None, key=0, mutation_mode=nt.MUT_ADAPTIVE, weights=None,
            chain_u=None, swap_u=None, trace=True, block_threads=0, chunks=None, flags=0, cluster_size=0):
    eng = GraphEngine(wl["n"], wl["ecap"], wl["temps"], wl["groups"], swap_frequency=swap_freq,
                      mutation_mode=mutation_mode, weights=weights, seed=seed, philox_key=key,
                      rng_mode=nt.RNG_INJECTED if chain_u is not None else nt.RNG_PHILOX,
                      block_threads=block_threads, flags=flags, cluster_size=cluster_size)
    try:
        eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
        init_cost = eng.download(positions=False, edges=False)["cost"]
        if chain_u is not None:
            eng.set_draws(chain_u, swap_u)
        passes = (iters + swap_freq - 1) // swap_freq if swap_freq > 0 else 0
        # without a step trace the swap decisions are still recorded (the step kernels then take the
        # paths they take in production: no per-step record is written)
        eng.enable_trace(iters if trace else 0, passes)
        for c in (chunks or [iters]):
            eng.run(c)
        eng.sync()
        res = eng.download()
        res["init_cost"] = init_cost
        res["stats"] = eng.stats()
        res["trace"], res["swap_trace"] = eng.read_trace(iters if trace else 0, passes)
        res["terms"] = eng.cost_terms()
        res["slots"] = eng.replica_slots()
        return res
    finally:
        eng.close()


def slots_after(swap_trace, T):
    """slot_of_replica after the recorded swap passes: replica t starts in slot t and an accepted
    swap of the pair (i, i+1) exchanges the replicas held by those two slots."""
    rep_of_slot = list(range(T))
    for row in swap_trace:
        for i in range(T - 1):
            if row[i]:
                rep_of_slot[i], rep_of_slot[i + 1] = rep_of_slot[i + 1], rep_of_slot[i]
    slot_of_rep = np.zeros(T, np.int32)
    for s, r in enumerate(rep_of_slot):
        slot_of_rep[r] = s
    return slot_of_rep


def assert_same_run(wl, gpu, orc, iters, *, trace=True):
    """Bit-exact comparison of a GPU run with the oracle's canonical-sum run."""
    T = wl["temps"]
    for g in range(wl["groups"]):
        o = orc[g]
        assert o["err"] == 0
        if trace:
            gt = gpu["trace"][:, g * T:(g + 1) * T]
            first = None
            for f in TRACE_FIELDS:
                if not np.array_equal(gt[f], o["trace"][f]):
                    bad = np.argwhere(gt[f] != o["trace"][f])[0]
                    if first is None or (bad[0], bad[1]) < (first[0], first[1]):
                        first = (bad[0], bad[1], f)
            if first is not None:
                s0, c0, f = first
                raise AssertionError(
                    f"ladder {g}: trace first differs at step {s0} chain {c0} (field {f!r}):\n"
                    f"  fields {TRACE_FIELDS}\n  gpu    {gt[s0, c0]}\n  oracle {o['trace'][s0, c0]}")
            assert np.array_equal(gpu["swap_trace"][:, g, :], o["swap_trace"]), f"ladder {g}: swap decisions differ"
        else:
            # no step trace on the GPU side: the oracle's own trace (when it has one) still gives the counts
            assert np.array_equal(gpu["swap_trace"][:, g, :], o["swap_trace"]), f"ladder {g}: swap decisions differ"
            assert np.array_equal(gpu["slots"][g], slots_after(o["swap_trace"], T)), f"ladder {g}: slot maps differ"
        assert np.array_equal(gpu["ecount"][g], o["ecount"]), f"ladder {g}: edge counts differ"
        assert np.array_equal(gpu["cost"][g], o["cost"]), f"ladder {g}: costs differ {gpu['cost'][g]} vs {o['cost']}"
        assert np.array_equal(gpu["pos"][g], o["pos"]), f"ladder {g}: positions differ"
        for t in range(T):
            ec = o["ecount"][t]
            assert np.array_equal(gpu["edges"][g, t, :ec], o["edges"][t, :ec]), f"ladder {g} chain {t}: edge lists differ"
    if all(o["trace"] is not None for o in orc):
        accepted = sum(int(((o["trace"]["flags"] & 4) != 0).sum()) for o in orc)
        assert gpu["stats"]["accepted"] == accepted, f"accepted {gpu['stats']['accepted']} vs oracle {accepted}"
    assert gpu["stats"]["swaps"] == sum(int(o["swap_trace"].sum()) for o in orc)
    assert gpu["stats"]["status_bits"] == 0, f"status bits {gpu['stats']['status_bits']:#x}"

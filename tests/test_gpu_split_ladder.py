"""One ladder split over two engines (what two ranks hold) against the same ladder in one
engine: run to each swap point, exchange the energies the way the all-gather would, run the
identical swap pass on both.  Everything must match the whole-ladder engine bit for bit."""
import numpy as np
import pytest
import torch

import parity_util as pu
from procgen_amd import native as nt
from procgen_amd import workloads as wlm
from procgen_amd.distributed import GraphEngineShard, SplitLadder
from procgen_amd.engine import GraphEngine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flags,cluster", [pytest.param(0, 0, id="incremental"), pytest.param(nt.FLAG_FULL_EVALUATION, 0, id="full"),
                                           pytest.param(nt.FLAG_STREAMING_INCREMENTAL, 1, id="streaming"),
                                           pytest.param(nt.FLAG_STREAMING_INCREMENTAL, 3, id="streaming-cluster")])
@pytest.mark.parametrize("n,Tl,parts,iters,f", [(60, 6, 2, 150, 9), (300, 8, 4, 120, 34), (20, 130, 2, 60, 4)])
def test_split_ladder_matches_whole_ladder(gpu_lib, n, Tl, parts, iters, f, flags, cluster):
    if cluster > 1 and Tl * cluster > 256:
        pytest.skip("that many workgroups cannot be resident at once: the engine refuses the configuration")
    wl = wlm.graph_workload("split-gpu", n, Tl, 1, edge_headroom=3.0, index=n)
    seed = f"split-{n}"
    whole = pu.gpu_run(wl, iters, f, seed=seed, trace=False, flags=flags, cluster_size=cluster)
    orc = pu.oracle_run(wl, iters, f, seed=seed, trace=False)
    pu.assert_same_run(wl, whole, orc, iters, trace=False)

    per = Tl // parts
    engines = []
    for p in range(parts):
        eng = GraphEngine(n, wl["ecap"], per, 1, swap_frequency=f, seed=seed, ladder_temp_count=Tl,
                          ladder_slot_base=p * per, flags=flags, cluster_size=cluster)
        lo = p * per
        eng.upload(wl["pos"][:, lo:lo + per], wl["edges"][:, lo:lo + per], wl["ecount"][:, lo:lo + per],
                   wl["temperatures"], wl["kind"], wl["value"])
        engines.append(eng)
    shards = [GraphEngineShard(e) for e in engines]
    all_e = torch.zeros(Tl, dtype=torch.float64, device="cuda")
    left = iters
    while left > 0:
        outs = [s.run_to_swap_point(left) for s in shards]
        assert len(set(outs)) == 1, "the shards disagree on the swap schedule"
        done, due = outs[0]
        left -= done
        if due:
            for p, s in enumerate(shards):              # stands in for the all-gather
                s.local_energies(all_e[p * per:(p + 1) * per])
            for s in shards:
                s.swap_pass(all_e)
    slots = [e.replica_slots()[0] for e in engines]
    for s in slots[1:]:
        assert np.array_equal(s, slots[0])
    assert np.array_equal(slots[0], whole["slots"][0]), "slot maps differ from the whole-ladder run"
    for p, eng in enumerate(engines):
        res = eng.download()
        st = eng.stats()
        assert st["status_bits"] == 0
        for t in range(per):
            s = slots[0][p * per + t]
            assert np.array_equal(res["pos"][0, t], whole["pos"][0, s])
            assert res["cost"][0, t] == whole["cost"][0, s]
            ec = whole["ecount"][0, s]
            assert res["ecount"][0, t] == ec and np.array_equal(res["edges"][0, t, :ec], whole["edges"][0, s, :ec])
        eng.close()


def test_single_rank_driver_equals_run(gpu_lib):
    """SplitLadder with a one-rank group is ParallelTempering.run itself."""
    n, Tl, iters, f = 80, 5, 90, 8
    wl = wlm.graph_workload("split-one", n, Tl, 1)
    whole = pu.gpu_run(wl, iters, f, seed="one", trace=False)
    eng = GraphEngine(n, wl["ecap"], Tl, 1, swap_frequency=f, seed="one", ladder_temp_count=Tl, ladder_slot_base=0)
    eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
    drv = SplitLadder(GraphEngineShard(eng))
    drv.run(iters)
    res = eng.download()
    eng.close()
    assert drv.gathers == (iters + f - 1) // f
    assert np.array_equal(res["cost"], whole["cost"]) and np.array_equal(res["pos"], whole["pos"])


def test_bench_split_ladder_rehearsal_two_ranks(gpu_lib):
    """bench.py --split-ladder as the driver launches it (torch.distributed.run, one process per rank), rehearsed
    with two ranks on this one GPU over gloo: one 8-temperature ladder, four per rank, an energy all-gather per
    swap interval."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PROCGEN_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(root, "bench.py"), "--gpus", "2", "--split-ladder", "--nodes", "3000",
           "--replicas", "4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["energy_gathers"] == rec["steps"] == 3
    assert rec["config"]["backend"].startswith("gloo") and rec["value"] > 0


@pytest.mark.parametrize("mode", ["ladders-per-rank", "split-ladder"])
def test_bench_starts_its_own_ranks(gpu_lib, mode):
    """Plain `python bench.py --gpus 2` (no launcher around it) starts two ranks itself and prints one line for the
    whole job; rehearsed on this one GPU over gloo."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PROCGEN_BENCH_REHEARSAL="1")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--nodes", "3000", "--replicas", "4", "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline"] + (["--split-ladder"] if mode == "split-ladder" else [])
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["value"] > 0
    assert rec["config"]["backend"].startswith("gloo")
    if mode == "split-ladder":
        assert rec["config"]["energy_gathers"] == 3

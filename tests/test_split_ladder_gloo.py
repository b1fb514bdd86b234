"""world_size-2 test of the split-ladder driver on CPUs (gloo): two ranks, each holding half
of one ladder as an oracle-backed shard, must reproduce a single-process run of the whole
ladder exactly (positions, edge lists, costs, swap decisions)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_binding as ob
import parity_util as pu
from oracle_shard import OracleShard
from procgen_amd import workloads as wlm
from procgen_amd.distributed import SplitLadder

N, TL, ITERS, F, SEED = 40, 4, 61, 7, "split-ladder"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _workload():
    wl = wlm.graph_workload("split", N, TL, 1, edge_headroom=3.0)
    wl["temperatures"][:] = [[6.0, 3.0, 1.5, 0.7]]
    return wl


def _rank_main(rank, world, port, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        wl = _workload()
        key = int(ob.lib().og_seed_key(SEED.encode()))
        per = TL // world
        lo = rank * per
        shard = OracleShard(wl["pos"][0, lo:lo + per], wl["edges"][0, lo:lo + per], wl["ecount"][0, lo:lo + per],
                            wl["kind"][0], wl["value"][0], wl["temperatures"][0], lo, F, key)
        drv = SplitLadder(shard, device=torch.device("cpu"))
        drv.run(ITERS // 2)
        drv.run(ITERS - ITERS // 2)             # a run may stop and resume anywhere in an interval
        np.savez(os.path.join(result_dir, f"rank{rank}.npz"), pos=shard.pos, edges=shard.edges, ecount=shard.ecount,
                 cost=shard.cost, slot_of_rep=shard.slot_of_rep, swaps=np.array(shard.swap_log), gathers=drv.gathers)
    finally:
        dist.destroy_process_group()


def test_two_ranks_reproduce_the_whole_ladder(tmp_path):
    world = 2
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    wl = _workload()
    full = pu.oracle_run(wl, ITERS, F, seed=SEED, trace=False)[0]
    passes = (ITERS + F - 1) // F
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert int(r0["gathers"]) == passes == int(r1["gathers"])
    assert np.array_equal(r0["slot_of_rep"], r1["slot_of_rep"]), "the ranks' slot maps diverged"
    assert np.array_equal(r0["swaps"], r1["swaps"])
    assert r0["swaps"].any() and not r0["swaps"].all()
    slot_of_rep = r0["slot_of_rep"]
    assert sorted(slot_of_rep.tolist()) == list(range(TL))
    per = TL // world
    for rank, res in enumerate((r0, r1)):
        for t in range(per):
            s = slot_of_rep[rank * per + t]                  # the temperature slot this state ended in
            assert np.array_equal(res["pos"][t], full["pos"][s])
            assert res["cost"][t] == full["cost"][s]
            ec = full["ecount"][s]
            assert res["ecount"][t] == ec and np.array_equal(res["edges"][t, :ec], full["edges"][s, :ec])

#!/usr/bin/env python3
"""Benchmark of the parallel-tempering hot path on MI355X.

Metric (BASELINE.json): Metropolis proposals/sec (whole node), 10k-node graph x 64 replicas.

  python bench.py --gpus N --steps K --warmup W

One "step" is one swap interval of the workload: f = swapFrequency(10000) = 200
iterations of ParallelTempering.run's loop on every replica (propose + full
weighted cost + Metropolis accept) followed by the sequential swap pass.  The
engine's default evaluation is incremental: the exact fixed-point cost sums are
updated by the terms the move changes, which yields bit for bit the cost a full
re-evaluation yields (--full-evaluation times the full re-evaluation kernels).  With N > 1 every rank (one process per GPU, launched by
torch.distributed.run) runs its own 64-replica ladder -- independent chains
shard across GPUs, so there is no data-path collective -- and the whole-job
value is the sum over ranks divided by the slowest rank's time ("weak").
All inputs are resident in HBM before the timed region.

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X FP32 vector spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nodes", type=int, default=10000)
    ap.add_argument("--replicas", type=int, default=64, help="temperatures per ladder")
    ap.add_argument("--ladders", type=int, default=1, help="independent ladders per GPU (configs[3]: --nodes 1000 --replicas 16 --ladders 512)")
    ap.add_argument("--layout", default="box", choices=["box", "sphere"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-evaluation", action="store_true", help="re-evaluate the full cost per proposal")
    ap.add_argument("--cluster", type=int, default=0, help="workgroups per replica of the full evaluation (0 = choose)")
    ap.add_argument("--workload", default="graphs", choices=["graphs", "chunks"],
                    help="graphs: BASELINE configs[1] (the headline); chunks: configs[2], 32^3 voxel chunks")
    ap.add_argument("--chunks", type=int, default=4096, help="chunks per GPU of the chunk workload")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    return ap.parse_args()


# PROCGEN_BENCH_REHEARSAL=1: every rank on cuda:0 with the gloo backend, to rehearse the N > 1 code path
# on a one-GPU box (the numbers then mean nothing).
REHEARSAL = os.environ.get("PROCGEN_BENCH_REHEARSAL") == "1"


def init_ranks(torch, dist, world, local_rank):
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if REHEARSAL:
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    return torch.cuda.current_device()


def cpu_baseline(wl, f):
    """Time the CPU oracle (a port of the reference's F# path; oracle/) on a bounded sample of
    the same workload: the reference's own algorithms and summation order, one thread per
    chain capped at the host's cores (reference src/Execution.fs:31-39)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    import parity_util as pu

    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16")), 16, wl["temps"])

    def timed(algo, sum_mode, iters):
        t0 = time.perf_counter()
        pu.oracle_run(wl, iters, f, seed="bench-cpu", algo=algo, sum_mode=sum_mode, trace=False, threads=cores)
        return time.perf_counter() - t0

    out = {}
    for name, algo, sum_mode in (("faithful", ob.ALGO_FAITHFUL, ob.SUM_FSHARP), ("optimised", ob.ALGO_OPTIMISED, ob.SUM_FSHARP)):
        t_init = timed(algo, sum_mode, 0)               # initial costs only
        iters, t = 1, timed(algo, sum_mode, 1)
        budget = ARGS.cpu_seconds / 2.0
        while t - t_init < budget / 4 and iters < 4096:  # grow the sample until it is long enough to time
            iters *= 4
            t = timed(algo, sum_mode, iters)
        out[name] = dict(rate=wl["temps"] * iters / max(t - t_init, 1e-9), iters=iters, seconds=t - t_init)
    return dict(value=out["faithful"]["rate"], unit="proposals/s", cores=cores, kind="port",
                sample=f"{wl['temps']} replicas x {out['faithful']['iters']} iterations of the 10k-node workload, "
                       f"reference algorithms (list-scan BFS, dictionary spatial hash), {out['faithful']['seconds']:.1f} s",
                optimised_value=out["optimised"]["rate"],
                optimised_sample=f"{wl['temps']} replicas x {out['optimised']['iters']} iterations, CSR BFS + grid hash, "
                                 f"{out['optimised']['seconds']:.1f} s")


def measured_traffic(kernel, record="pmc_traffic.json"):
    """HBM bytes per launch of the step kernel from the committed PMC passes (rocprofv3 cannot
    run inside this process); None when no measurement of this kernel is on file."""
    path = os.path.join(ROOT, "profiles", record)
    try:
        with open(path) as fh:
            rec = json.load(fh)
        if rec.get("kernel") == kernel:
            return rec["hbm_bytes_per_launch"], rec.get("note", "")
    except (OSError, ValueError, KeyError):
        pass
    return None, "no PMC measurement of this kernel on file"


def chunk_cpu_baseline(wl, f):
    """The terrain oracle (reference summation order) on one ladder for a few iterations."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    import terrain_binding as tb

    cfg = tb.default_config(ob.SUM_FSHARP)
    T = wl["temps"]
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16")), 16, T)

    def timed(iters):
        t0 = time.perf_counter()
        tb.run(wl["density"][0], wl["moisture"][0], wl["temperature"][0], wl["material"][0], wl["temperatures"][0],
               iters, f, cfg, key=1, trace=False, threads=cores)
        return time.perf_counter() - t0

    t_init = timed(0)
    iters, t = 8, timed(8)
    while t - t_init < ARGS.cpu_seconds / 4 and iters < 65536:
        iters *= 4
        t = timed(iters)
    return dict(value=T * iters / max(t - t_init, 1e-9), unit="proposals/s", cores=cores, kind="port",
                sample=f"one chunk ladder ({T} temperatures) x {iters} iterations, full terrain cost per proposal, "
                       f"{t - t_init:.1f} s")


def main_chunks():
    """BASELINE configs[2]: 32^3 voxel terrain chunks, 6 temperatures each, batched."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import numpy as np
    import torch
    import torch.distributed as dist

    dev = init_ranks(torch, dist, world, local_rank)
    from procgen_amd import workloads as wlm
    from procgen_amd.engine import ChunkEngine

    G, T = ARGS.chunks, 6
    distinct = min(G, 32)
    wl = wlm.chunk_workload("c3", distinct, T, index=rank)
    rep = (G + distinct - 1) // distinct
    tile = lambda a: np.tile(a, (rep,) + (1,) * (a.ndim - 1))[:G]       # noqa: E731  (distinct chunks repeated; every ladder has its own draws)
    eng = ChunkEngine(T, G, seed="procgen-bench/c3", device=dev, group_base=rank * G)
    f = eng.swap_frequency
    eng.upload(tile(wl["density"]), tile(wl["moisture"]), tile(wl["temperature"]), tile(wl["material"]), tile(wl["temperatures"]))
    eng.run(1)
    for _ in range(ARGS.warmup):
        eng.run(f)
    eng.sync()
    eng.kernel_time(reset=True)
    st0 = eng.stats()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(ARGS.steps):
        eng.run(f)
    eng.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if REHEARSAL else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kms, klaunch = eng.kernel_time()
    st1 = eng.stats()
    props = st1["proposals"] - st0["proposals"]
    assert props == ARGS.steps * f * G * T and st1["status_bits"] == 0
    if rank == 0:
        V = wl["V"]
        bpp = wlm.chunk_algorithmic_bytes_per_proposal(V)
        per_launch = props / max(klaunch, 1)
        launch_s = kms / max(klaunch, 1) * 1e-3
        achieved = bpp * per_launch / launch_s / 1e9
        traffic, traffic_note = measured_traffic("pt_chunk_steps_kernel" if G == 4096 else None, "pmc_traffic_chunks.json")
        line = {
            "metric": "Metropolis proposals/sec (whole node), 32^3 voxel chunks x 6 temperatures", "value": props * world / elapsed,
            "unit": "proposals/s", "n_gpus": world, "steps": ARGS.steps, "warmup": ARGS.warmup,
            "ms_per_step": elapsed / ARGS.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 voxel fields, f64 cost, i128 fixed-point sums", "data": "synthetic",
            "config": {"workload": f"configs[2]: {G} voxel chunks (32^3) x {T} temperatures per GPU, terrain constraints, "
                                   f"ladder 0.1-5.0, swap every {f}, incremental exact cost update per proposal",
                       "chunks_per_gpu": G, "temperatures": T, "swap_frequency": f,
                       "accept_rate": (st1["accepted"] - st0["accepted"]) / max(props, 1)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_note": traffic_note, "kernel": "pt_chunk_steps_kernel",
                         "kernel_ms_avg": kms / max(klaunch, 1),
                         "launches": klaunch, "proposals_per_launch": per_launch, "algorithmic_bytes_per_proposal": bpp,
                         "algorithmic_note": "bytes one FULL terrain-cost evaluation touches (the reference's work per proposal)"},
        }
        if not ARGS.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = chunk_cpu_baseline(wl, f)
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist

    dev = init_ranks(torch, dist, world, local_rank)

    import numpy as np
    from procgen_amd import native as nt
    from procgen_amd import workloads as wlm
    from procgen_amd.engine import GraphEngine

    n, R = ARGS.nodes, ARGS.replicas
    G = ARGS.ladders
    distinct = min(G, 16)                        # distinct ladders generated; tiled when more are asked for
    wl = wlm.graph_workload("c2", n, R, distinct, layout=ARGS.layout, index=rank)
    if G > distinct:
        reps = (G + distinct - 1) // distinct
        for k in ("pos", "edges", "ecount", "temperatures", "kind", "value"):
            wl[k] = np.ascontiguousarray(np.tile(wl[k], (reps,) + (1,) * (wl[k].ndim - 1))[:G])
        wl["groups"] = G
    eng = GraphEngine(n, wl["ecap"], R, G, seed="procgen-bench/c2", device=dev, group_base=rank * G,
                      flags=nt.FLAG_FULL_EVALUATION if ARGS.full_evaluation else 0, cluster_size=ARGS.cluster)
    info = eng.info()
    f = eng.swap_frequency
    eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
    eng.run(1)                                  # step 0 and its swap pass: intervals then start at 1, f+1, ...
    for _ in range(ARGS.warmup):
        eng.run(f)
    eng.sync()
    eng.kernel_time(reset=True)
    st0 = eng.stats()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    barrier()
    t0 = time.perf_counter()
    for _ in range(ARGS.steps):
        eng.run(f)
    eng.sync()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if REHEARSAL else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    kms, klaunch = eng.kernel_time()
    st1 = eng.stats()
    res = eng.download(positions=False, edges=True)
    proposals_rank = st1["proposals"] - st0["proposals"]
    assert proposals_rank == ARGS.steps * f * R * G, (proposals_rank, ARGS.steps * f * R * G)
    assert st1["status_bits"] == 0, f"device status bits {st1['status_bits']:#x}"
    total = proposals_rank * world
    value = total / elapsed

    if rank == 0:
        e_mean = float(res["ecount"].mean())
        bytes_per_prop = wlm.algorithmic_bytes_per_proposal(n, e_mean)
        flops_per_prop = wlm.algorithmic_flops_per_proposal(n, e_mean)
        props_per_launch = proposals_rank / max(klaunch, 1)
        launch_s = (kms / max(klaunch, 1)) * 1e-3
        achieved = bytes_per_prop * props_per_launch / launch_s / 1e9
        valu = flops_per_prop * props_per_launch / launch_s / 1e12
        if info["incremental"]:
            kernel = ("pt_graph_delta_kernel" if info["lds_placement"]
                      else "pt_graph_bigc_kernel" if info["cluster_size"] > 1 else "pt_graph_big_kernel")
            how = "incremental exact cost update per proposal (same bits as a full re-evaluation)"
            if not info["lds_placement"] and info["cluster_size"] > 1:
                how += f", {info['cluster_size']} workgroups per replica"
        elif info["cluster_size"] > 1:
            kernel, how = "pt_graph_cluster_kernel", f"full cost re-evaluation per proposal by {info['cluster_size']} workgroups per replica"
        else:
            kernel, how = "pt_graph_steps_kernel", "full cost re-evaluation per proposal"
        # the PMC record on file was taken at the headline shape only
        if (n, R, G) == (10000, 64, 1):
            traffic, traffic_note = measured_traffic(kernel)
        elif (n, R, G) == (1000000, 16, 1):
            traffic, traffic_note = measured_traffic(kernel, "pmc_traffic_1m.json")
        else:
            traffic, traffic_note = measured_traffic(None)
        measured_valu = None
        if (n, R, G) == (10000, 64, 1) and kernel == "pt_graph_delta_kernel":
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_sq.json")) as fh:
                    sq = json.load(fh)["counters"]
                lane_instr = sq["SQ_INSTS_VALU"] * 64.0 / launch_s         # VALU lane-instructions per second, this run's launch time
                # peak issue: one lane-instruction per lane per clock = half the FMA flop peak
                measured_valu = {"valu_wave_instructions_per_launch": sq["SQ_INSTS_VALU"],
                                 "salu_wave_instructions_per_launch": sq["SQ_INSTS_SALU"],
                                 "lds_wave_instructions_per_launch": sq["SQ_INSTS_LDS"],
                                 "lane_instructions_per_s": lane_instr,
                                 "frac_of_peak_issue": lane_instr / (FP32_VALU_PEAK_TFLOPS * 1e12 / 2.0),
                                 "note": "rocprofv3 --pmc SQ_INSTS_* of one launch (profiles/pmc_sq.json) over this run's launch "
                                         "time; 64 of 256 CUs hold a replica"}
            except (OSError, ValueError, KeyError):
                pass
        line = {
            "metric": "Metropolis proposals/sec (whole node), 10k-node graph x 64 replicas" if (n, R, G) == (10000, 64, 1)
                      else f"Metropolis proposals/sec (whole node), {n}-node graph x {R * G} replicas",
            "value": value, "unit": "proposals/s", "n_gpus": world, "steps": ARGS.steps, "warmup": ARGS.warmup,
            "ms_per_step": elapsed / ARGS.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 positions, f64 cost, i128 fixed-point sums", "data": "synthetic",
            "config": {"workload": f"{'configs[1]' if (n, R, G) == (10000, 64, 1) else 'other shape'}: {n}-node spatial graph x {R} replicas"
                                   f"{'' if G == 1 else f' x {G} ladders'} per GPU, E~1.5N, "
                                   f"uniform-box layout, ladder 0.5-5.0, swap every {f}, {how}",
                       "nodes": n, "replicas_per_gpu": R * G, "ladders_per_gpu": G, "edges_mean": e_mean, "swap_frequency": f,
                       "iterations_per_step": f, "accept_rate": (st1["accepted"] - st0["accepted"]) / max(proposals_rank, 1),
                       "swap_rate": st1["swaps"] / max(st1["swap_attempts"], 1)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": kernel, "kernel_ms_avg": kms / max(klaunch, 1), "launches": klaunch,
                         "proposals_per_launch": props_per_launch,
                         "algorithmic_bytes_per_proposal": bytes_per_prop,
                         "algorithmic_note": "bytes and lane-ops one FULL cost evaluation touches (the reference's work per "
                                             "proposal); the engine may do less work for the same result",
                         "lds_bytes_per_workgroup": info["lds_bytes"], "block_threads": info["block_threads"],
                         "valu": {"achieved": valu, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": valu / FP32_VALU_PEAK_TFLOPS, "lane_ops_per_proposal": flops_per_prop,
                                  "measured": measured_valu}},
        }
        if not ARGS.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(wl, f)
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    ARGS = parse()
    if ARGS.workload == "chunks":
        main_chunks()
    else:
        main()

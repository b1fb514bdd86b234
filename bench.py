#!/usr/bin/env python3
"""Benchmark of the parallel-tempering hot path on MI355X.

Metric (BASELINE.json): Metropolis proposals/sec (whole node), 10k-node graph x 64 replicas.

  python bench.py --gpus N --steps K --warmup W

One "step" is one swap interval of the workload: f = swapFrequency(10000) = 200 iterations of
ParallelTempering.run's loop on every replica (propose + weighted cost + Metropolis accept) followed by
the sequential swap pass.  The engine's default evaluation is incremental: the exact fixed-point cost
sums are updated by the terms the move changes, which yields bit for bit the cost a full re-evaluation
yields (--full-evaluation times the full re-evaluation kernels).  Without --steps the timed region is
sized to last about 0.6 s.

Multi-GPU: one process per GPU.  Started under torch.distributed.run (RANK / WORLD_SIZE in the environment) this
process is one rank; started plainly as `python bench.py --gpus N` with N > 1 it starts the N ranks itself
(a child `python -m torch.distributed.run --nproc-per-node N bench.py ...`, before anything here touches the
GPU) and passes on their output and exit code.
  default          every rank runs its own ladder: independent chains shard across GPUs with no
                   data-path collective; the whole-job value is the sum over ranks divided by the slowest
                   rank's time ("weak").
  --split-ladder   ONE ladder of 16 x N temperatures over N ranks (BASELINE configs[4] at N = 8: 10^6 nodes,
                   128 replicas): every swap interval ends with an all-gather of the per-replica energies
                   (RCCL over xGMI) and the identical swap pass on every rank; states never move.
All inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X FP32 vector spec peak
TIMED_SECONDS = 0.6            # length of the timed region when --steps is not given


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed swap intervals (0: as many as fill about 0.6 s)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nodes", type=int, default=0, help="nodes per graph (default 10000; 1000000 with --split-ladder)")
    ap.add_argument("--replicas", type=int, default=0, help="temperatures per ladder held by one GPU (default 64; 16 with --split-ladder)")
    ap.add_argument("--ladders", type=int, default=1, help="independent ladders per GPU (configs[3]: --nodes 1000 --replicas 16 --ladders 512)")
    ap.add_argument("--layout", default="box", choices=["box", "sphere"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-evaluation", action="store_true", help="re-evaluate the full cost per proposal")
    ap.add_argument("--cluster", type=int, default=0, help="workgroups per replica (0 = choose)")
    ap.add_argument("--block-threads", type=int, default=0, help="threads per workgroup (0 = choose)")
    ap.add_argument("--split-ladder", action="store_true",
                    help="one ladder of replicas x N temperatures split over the N ranks, energies all-gathered per swap interval")
    ap.add_argument("--workload", default="graphs", choices=["graphs", "chunks"],
                    help="graphs: BASELINE configs[1] (the headline); chunks: configs[2], 32^3 voxel chunks")
    ap.add_argument("--chunks", type=int, default=4096, help="chunks per GPU of the chunk workload")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    a = ap.parse_args()
    if not a.nodes:
        a.nodes = 1000000 if a.split_ladder else 10000
    if not a.replicas:
        a.replicas = 16 if a.split_ladder else 64
    return a


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


def host_cores(limit):
    return max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16")), 16, limit))


def cpu_baseline(wl, f):
    """Time the CPU oracle (a port of the reference's F# path; oracle/) on a bounded sample of the same
    workload: the reference's own algorithms and summation order, one thread per chain capped at the
    host's cores (reference src/Execution.fs:31-39).  The reference's list-scan search is O(V E) per
    proposal, so beyond 50 000 nodes only the variant with CSR search and grid hash is timed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    import parity_util as pu

    T, n = wl["temps"], wl["n"]
    one = dict(wl)                                  # one ladder is sample enough
    for k in ("pos", "edges", "ecount", "temperatures", "kind", "value"):
        one[k] = wl[k][:1]
    one["groups"] = 1
    one["temperatures"] = one["temperatures"][:, :T]
    cores = host_cores(T)

    def timed(algo, sum_mode, iters):
        t0 = time.perf_counter()
        pu.oracle_run(one, iters, f, seed="bench-cpu", algo=algo, sum_mode=sum_mode, trace=False, threads=cores)
        return time.perf_counter() - t0

    modes = [("optimised", ob.ALGO_OPTIMISED, ob.SUM_FSHARP)]
    if n <= 50000:
        modes.insert(0, ("faithful", ob.ALGO_FAITHFUL, ob.SUM_FSHARP))
    out = {}
    for name, algo, sum_mode in modes:
        t_init = timed(algo, sum_mode, 0)               # initial costs only
        iters, t = 1, timed(algo, sum_mode, 1)
        budget = ARGS.cpu_seconds / len(modes)
        while t - t_init < budget / 4 and iters < 4096 and t < budget:      # grow the sample until it is long enough to time
            iters *= 4
            t = timed(algo, sum_mode, iters)
        out[name] = dict(rate=T * iters / max(t - t_init, 1e-9), iters=iters, seconds=t - t_init)
    opt = out["optimised"]
    rec = dict(unit="proposals/s", cores=cores, kind="port", optimised_value=opt["rate"],
               optimised_sample=f"{T} replicas x {opt['iters']} iterations of the {n}-node workload, CSR search + grid hash, "
                                f"{opt['seconds']:.1f} s")
    if "faithful" in out:
        fa = out["faithful"]
        rec.update(value=fa["rate"],
                   sample=f"{T} replicas x {fa['iters']} iterations of the {n}-node workload, reference algorithms "
                          f"(list-scan BFS, dictionary spatial hash), {fa['seconds']:.1f} s")
    else:
        rec.update(value=opt["rate"],
                   sample=rec["optimised_sample"] + " (the reference's own list-scan search is O(V E) per proposal: "
                          "not timed at this size; this figure flatters the CPU)")
    return rec


def pmc_record(name, kernel):
    """A committed PMC measurement (rocprofv3 cannot run inside this process): HBM bytes per proposal of
    `kernel`, or None when none is on file."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            rec = json.load(fh)
        if rec.get("kernel") == kernel and rec.get("proposals_per_launch"):
            return rec["hbm_bytes_per_launch"] / rec["proposals_per_launch"], rec.get("note", "")
    except (OSError, ValueError, KeyError):
        pass
    return None, "no PMC measurement of this kernel and shape on file"


def chunk_cpu_baseline(wl, f):
    """The terrain oracle (reference summation order) on one ladder for a few iterations."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    import terrain_binding as tb

    cfg = tb.default_config(ob.SUM_FSHARP)
    T = wl["temps"]
    cores = host_cores(T)

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


class Timed:
    """Warm-up, then K steps bracketed by barrier + synchronize on both sides; MAX over ranks."""

    def __init__(self, torch, dist, world, step_fn, sync_fn):
        self.torch, self.dist, self.world, self.step_fn, self.sync_fn = torch, dist, world, step_fn, sync_fn

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()
        self.sync_fn()

    def max_over_ranks(self, x):
        if self.world > 1:
            tt = self.torch.tensor([x], dtype=self.torch.float64, device="cpu" if REHEARSAL else "cuda")
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            return float(tt.item())
        return x

    def run(self, before_timed):
        for _ in range(ARGS.warmup):
            self.step_fn()
        self.sync_fn()
        steps = ARGS.steps
        if steps <= 0:                              # size the timed region: one more step, timed
            self.barrier()
            t0 = time.perf_counter()
            self.step_fn()
            self.sync_fn()
            one = self.max_over_ranks(time.perf_counter() - t0)
            steps = int(min(5000, max(2, math.ceil(TIMED_SECONDS / max(one, 1e-6)))))
        before_timed()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step_fn()
        self.sync_fn()
        self.torch.cuda.synchronize()
        t1 = time.perf_counter()
        self.barrier()
        return steps, self.max_over_ranks(t1 - t0)


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
    st0 = {}

    def before_timed():
        eng.kernel_time(reset=True)
        st0.update(eng.stats())

    steps, elapsed = Timed(torch, dist, world, lambda: eng.run(f), eng.sync).run(before_timed)
    kms, klaunch = eng.kernel_time()
    st1 = eng.stats()
    props = st1["proposals"] - st0["proposals"]
    assert props == steps * f * G * T and st1["status_bits"] == 0
    if rank == 0:
        V = wl["V"]
        per_launch = props / max(klaunch, 1)
        launch_s = kms / max(klaunch, 1) * 1e-3
        traffic, traffic_note = pmc_record("pmc_traffic_chunks.json", "pt_chunk_steps_kernel")
        ref_bpp = wlm.chunk_algorithmic_bytes_per_proposal(V)
        # bytes one incremental proposal must move: the voxel's record and its six neighbours' (16 B each),
        # the replica's header, one record written back
        alg_bpp = 7 * 16 + 64 + 16
        bpp = traffic if traffic is not None else alg_bpp
        achieved = bpp * per_launch / launch_s / 1e9
        line = {
            "metric": "Metropolis proposals/sec (whole node), 32^3 voxel chunks x 6 temperatures", "value": props * world / elapsed,
            "unit": "proposals/s", "n_gpus": world, "steps": steps, "warmup": ARGS.warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 voxel fields, f64 cost, i128 fixed-point sums", "data": "synthetic",
            "config": {"workload": f"configs[2]: {G} voxel chunks (32^3) x {T} temperatures per GPU, terrain constraints, "
                                   f"ladder 0.1-5.0, swap every {f}, incremental exact cost update per proposal",
                       "chunks_per_gpu": G, "temperatures": T, "swap_frequency": f,
                       "accept_rate": (st1["accepted"] - st0["accepted"]) / max(props, 1)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if traffic is None else traffic * per_launch,
                         "traffic_note": traffic_note, "kernel": "pt_chunk_steps_kernel",
                         "bytes_per_proposal": bpp,
                         "bytes_note": "HBM bytes per proposal: the committed PMC measurement of this kernel when one is on file, "
                                       "else what one incremental proposal must move (7 voxel records, header, one write)",
                         "kernel_ms_avg": kms / max(klaunch, 1), "launches": klaunch, "proposals_per_launch": per_launch,
                         "reference_work": {"bytes_per_proposal": ref_bpp, "rate_GBs": ref_bpp * per_launch / launch_s / 1e9,
                                            "note": "bytes one FULL terrain-cost evaluation touches (the reference's work per "
                                                    "proposal) times the proposal rate: a throughput in units of reference work, "
                                                    "not bytes moved"}},
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
    from procgen_amd.distributed import GraphEngineShard, SplitLadder
    from procgen_amd.engine import GraphEngine

    n, R = ARGS.nodes, ARGS.replicas
    G = ARGS.ladders
    split = ARGS.split_ladder
    flags = nt.FLAG_FULL_EVALUATION if ARGS.full_evaluation else 0
    if split:
        # one ladder of R x world temperatures; this rank holds replicas [rank R, (rank + 1) R)
        assert G == 1, "--split-ladder runs one ladder"
        Tl = R * world
        whole = wlm.graph_workload("c5", n, Tl, 1, layout=ARGS.layout)
        wl = dict(whole)
        for k in ("pos", "edges", "ecount"):
            wl[k] = np.ascontiguousarray(whole[k][:, rank * R:(rank + 1) * R])
        wl["temps"] = R
        eng = GraphEngine(n, wl["ecap"], R, 1, seed="procgen-bench/c5", device=dev, ladder_temp_count=Tl,
                          ladder_slot_base=rank * R, flags=flags, cluster_size=ARGS.cluster,
                          block_threads=ARGS.block_threads)
        driver = SplitLadder(GraphEngineShard(eng), device=torch.device("cpu") if REHEARSAL else None)
    else:
        distinct = min(G, 16)                        # distinct ladders generated; tiled when more are asked for
        wl = wlm.graph_workload("c2", n, R, distinct, layout=ARGS.layout, index=rank)
        if G > distinct:
            reps = (G + distinct - 1) // distinct
            for k in ("pos", "edges", "ecount", "temperatures", "kind", "value"):
                wl[k] = np.ascontiguousarray(np.tile(wl[k], (reps,) + (1,) * (wl[k].ndim - 1))[:G])
            wl["groups"] = G
        eng = GraphEngine(n, wl["ecap"], R, G, seed="procgen-bench/c2", device=dev, group_base=rank * G,
                          flags=flags, cluster_size=ARGS.cluster, block_threads=ARGS.block_threads)
        driver = None
    info = eng.info()
    f = eng.swap_frequency
    eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
    step_fn = (lambda: driver.run(f)) if split else (lambda: eng.run(f))
    (driver.run if split else eng.run)(1)        # step 0 and its swap pass: intervals then start at 1, f+1, ...
    st0, g0 = {}, [0]

    def before_timed():
        eng.kernel_time(reset=True)
        st0.update(eng.stats())
        g0[0] = driver.gathers if split else 0

    steps, elapsed = Timed(torch, dist, world, step_fn, eng.sync).run(before_timed)

    kms, klaunch = eng.kernel_time()
    st1 = eng.stats()
    res = eng.download(positions=False, edges=True)
    proposals_rank = st1["proposals"] - st0["proposals"]
    assert proposals_rank == steps * f * R * G, (proposals_rank, steps * f * R * G)
    assert st1["status_bits"] == 0, f"device status bits {st1['status_bits']:#x}"
    gathers = (driver.gathers - g0[0]) if split else 0
    assert not split or gathers == steps, (gathers, steps)
    total = proposals_rank * world
    value = total / elapsed

    if rank == 0:
        e_mean = float(res["ecount"].mean())
        ref_bytes = wlm.algorithmic_bytes_per_proposal(n, e_mean)
        ref_flops = wlm.algorithmic_flops_per_proposal(n, e_mean)
        props_per_launch = proposals_rank / max(klaunch, 1)
        launch_s = (kms / max(klaunch, 1)) * 1e-3
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
        headline = (n, R, G) == (10000, 64, 1) and not split
        traffic_pp, traffic_note = (pmc_record("pmc_traffic.json", kernel) if headline else
                                    pmc_record("pmc_traffic_1m.json", kernel) if (n, R, G) == (1000000, 16, 1) else
                                    (None, "no PMC measurement of this kernel and shape on file"))
        valu = ref_flops * props_per_launch / launch_s / 1e12
        hbm_equiv = ref_bytes * props_per_launch / launch_s / 1e9
        common = {"kernel": kernel, "kernel_ms_avg": kms / max(klaunch, 1), "launches": klaunch,
                  "proposals_per_launch": props_per_launch, "lds_bytes_per_workgroup": info["lds_bytes"],
                  "block_threads": info["block_threads"],
                  "traffic": None if traffic_pp is None else traffic_pp * props_per_launch, "traffic_note": traffic_note}
        if info["lds_placement"]:
            # replica state resident in LDS: HBM sees next to nothing, the bound is vector issue
            measured_valu = None
            if headline and kernel == "pt_graph_delta_kernel":
                try:
                    with open(os.path.join(ROOT, "profiles", "pmc_sq.json")) as fh:
                        sq = json.load(fh)["counters"]
                    lane_instr = sq["SQ_INSTS_VALU"] * 64.0 / launch_s         # VALU lane-instructions per second, this run's launch time
                    measured_valu = {"valu_wave_instructions_per_launch": sq["SQ_INSTS_VALU"],
                                     "salu_wave_instructions_per_launch": sq["SQ_INSTS_SALU"],
                                     "lds_wave_instructions_per_launch": sq["SQ_INSTS_LDS"],
                                     "lane_instructions_per_s": lane_instr,
                                     # peak issue: one lane-instruction per lane per clock = half the FMA flop peak
                                     "frac_of_peak_issue": lane_instr / (FP32_VALU_PEAK_TFLOPS * 1e12 / 2.0),
                                     "note": "rocprofv3 --pmc SQ_INSTS_* of one launch (profiles/pmc_sq.json) over this run's "
                                             "launch time; 64 of 256 CUs hold a replica"}
                    # Share of the time in which the SIMDs that hold a replica have a vector instruction in flight, from the
                    # counters alone: SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES both count quad-cycles, the second summed over
                    # the waves, so their ratio is one wave's share and `waves_per_simd` of them share a SIMD.
                    waves_per_simd = info["block_threads"] / 64.0 / 4.0
                    if sq.get("SQ_WAVE_CYCLES"):
                        measured_valu["valu_busy_on_simds_in_use"] = {
                            "frac": sq["SQ_ACTIVE_INST_VALU"] * waves_per_simd / sq["SQ_WAVE_CYCLES"],
                            "waves_per_simd": waves_per_simd,
                            "quad_cycles_per_valu_instruction": sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_INSTS_VALU"],
                            "note": "SQ_ACTIVE_INST_VALU x waves per SIMD / SQ_WAVE_CYCLES (profiles/pmc_sq.json; clock-free: "
                                    "both are quad-cycle counts).  The counter books one quad-cycle per instruction whatever it "
                                    "is; a saturated SIMD retires this kernel's mix (packed fp32, fp64, conversions, DPP) more "
                                    "slowly than that (profiles/r03_instr_rate.txt), so the SIMDs are fuller than this figure "
                                    "says and emptier than the instruction-rate estimate: the serial stretches of a step "
                                    "(one wave at work) are where the empty slots sit"}
                except (OSError, ValueError, KeyError):
                    pass
            roofline = dict(common, bound="valu", achieved=valu, peak=FP32_VALU_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=valu / FP32_VALU_PEAK_TFLOPS, lane_ops_per_proposal=ref_flops,
                            algorithmic_note="lane-ops one FULL cost evaluation needs (the reference's work per proposal: "
                                             "11E + 30N + (27 + 10 k)N + 2E) times the proposal rate, against the fp32 vector "
                                             "peak; the incremental kernel does less work for the same result",
                            measured=measured_valu,
                            hbm_equivalent={"rate_GBs": hbm_equiv, "bytes_per_proposal": ref_bytes,
                                            "note": "bytes one full evaluation touches times the proposal rate (reference work, "
                                                    "not bytes moved: the state is in LDS)"})
        else:
            # replica state streamed from HBM
            moved = n * 20                              # one perturbation: positions read, distances written and read back
            bpp = traffic_pp if traffic_pp is not None else moved
            achieved = bpp * props_per_launch / launch_s / 1e9
            roofline = dict(common, bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                            bytes_per_proposal=bpp,
                            bytes_note="HBM bytes per proposal: the committed PMC measurement of this kernel and shape when one is "
                                       "on file, else what one incremental perturbation streams at most (12N + 8N)",
                            reference_work={"bytes_per_proposal": ref_bytes, "rate_GBs": hbm_equiv,
                                            "lane_ops_rate_TFLOPs": valu})
        shape = ("configs[1]" if headline else "configs[4] shape" if split else "other shape")
        line = {
            "metric": "Metropolis proposals/sec (whole node), 10k-node graph x 64 replicas" if headline
                      else f"Metropolis proposals/sec (whole node), {n}-node graph x {R * G * (world if split else 1)} replicas"
                           f"{' (one ladder split over the GPUs)' if split else ''}",
            "value": value, "unit": "proposals/s", "n_gpus": world, "steps": steps, "warmup": ARGS.warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 positions, f64 cost, i128 fixed-point sums", "data": "synthetic",
            "config": {"workload": f"{shape}: {n}-node spatial graph x {R} replicas"
                                   f"{'' if G == 1 else f' x {G} ladders'} per GPU, E~1.5N, "
                                   f"uniform-box layout, ladder 0.5-5.0"
                                   f"{f' of {R * world} temperatures split over {world} ranks, energies all-gathered per swap interval' if split else ''}"
                                   f", swap every {f}, {how}",
                       "nodes": n, "replicas_per_gpu": R * G, "ladders_per_gpu": G, "edges_mean": e_mean, "swap_frequency": f,
                       "iterations_per_step": f, "accept_rate": (st1["accepted"] - st0["accepted"]) / max(proposals_rank, 1),
                       "swap_rate": st1["swaps"] / max(st1["swap_attempts"], 1),
                       "energy_gathers": gathers if split else None,
                       "backend": ("gloo (rehearsal)" if REHEARSAL else "nccl") if world > 1 else None},
            "roofline": roofline,
        }
        if not ARGS.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(wl, f)
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def spawn_ranks(n):
    """`python bench.py --gpus N` outside a launcher: run the N ranks as a child process and relay the result."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


if __name__ == "__main__":
    ARGS = parse()
    if ARGS.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(ARGS.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != max(ARGS.gpus, 1):
        sys.exit(f"bench.py: --gpus {ARGS.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')} ranks")
    if ARGS.workload == "chunks":
        main_chunks()
    else:
        main()

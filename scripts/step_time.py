#!/usr/bin/env python3
"""Kernel time per proposal of the graph engine for one mutation mix (diagnostic).
usage: step_time.py [n] [replicas] [iters] [perturb_only 0/1] [flags] [tmin] [tmax]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from procgen_amd import native as nt  # noqa: E402
from procgen_amd import workloads as wlm  # noqa: E402
from procgen_amd.engine import GraphEngine  # noqa: E402

a = sys.argv[1:]
n = int(a[0]) if len(a) > 0 else 10000
R = int(a[1]) if len(a) > 1 else 64
iters = int(a[2]) if len(a) > 2 else 400
ponly = int(a[3]) if len(a) > 3 else 0
flags = int(a[4]) if len(a) > 4 else 0
tmin = float(a[5]) if len(a) > 5 else 0.5
tmax = float(a[6]) if len(a) > 6 else 5.0
wl = wlm.graph_workload("c2", n, R, 1, t_min=tmin, t_max=tmax)
eng = GraphEngine(n, wl["ecap"], R, 1, seed="procgen-bench/c2", flags=flags, block_threads=int(os.environ.get("BLOCK", "0")),
                  mutation_mode=nt.MUT_PERTURB_ONLY if ponly else nt.MUT_ADAPTIVE)
eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
eng.run(1 + eng.swap_frequency)
eng.sync()
eng.kernel_time(reset=True)
eng.run(iters)
eng.sync()
ms, launches = eng.kernel_time()
st = eng.stats()
print(f"n={n} R={R} perturb_only={ponly} T=[{tmin},{tmax}] info={eng.info()} kernel_ms={ms:.3f} "
      f"us/step={ms * 1e3 / iters:.2f} proposals/s={R * iters / (ms * 1e-3):.0f} accept={st['accepted'] / st['proposals']:.3f}")
eng.close()

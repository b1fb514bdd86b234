#!/usr/bin/env python3
"""collision_detection (host arrays in, sorted pairs out) and pt_collision_detect_device (device arrays) against the
CPU routine's restatement, milliseconds per call (diagnostic).  usage: collision_time.py [n]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob                 # noqa: E402
from procgen_amd import native as nt        # noqa: E402
from procgen_amd import workloads as wlm    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
import torch                                # noqa: E402
L = nt.load()
pos = wlm.box_positions(n, np.random.default_rng(1))
p64 = pos.astype(np.float64).reshape(-1)
pairs = np.zeros(max(n * n // 2, 1) * 2, np.int32)
count = np.zeros(1, np.int32)
for i in range(13):
    if i == 3:
        t0 = time.perf_counter()
    L.collision_detection(nt.ptr(p64), nt.ptr(pairs), n, 0.5, nt.ptr(count))
t_host = (time.perf_counter() - t0) / 10
d_pos = torch.from_numpy(pos).cuda()
d_pairs = torch.zeros((int(count[0]) + 16, 2), dtype=torch.int32, device="cuda")
d_count = torch.zeros(1, dtype=torch.int32, device="cuda")
for i in range(23):
    if i == 3:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    nt.check(L.pt_collision_detect_device(C.c_void_p(d_pos.data_ptr()), n, 0.5, C.c_void_p(d_pairs.data_ptr()), d_pairs.shape[0],
                                          C.c_void_p(d_count.data_ptr()), None))
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / 20
t0 = time.perf_counter()
want = ob.collision_pairs(pos, 0.5)
t_cpu = time.perf_counter() - t0
assert int(d_count.cpu()[0]) == len(want) == int(count[0])
print(f"n={n}: {len(want)} pairs; collision_detection {t_host * 1e3:.3f} ms per call (host arrays), pt_collision_detect_device "
      f"{t_dev * 1e3:.3f} ms per call (device arrays), CPU routine (grid restatement, one thread) {t_cpu * 1e3:.1f} ms")

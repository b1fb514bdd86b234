"""What speculative evaluation of consecutive proposals could buy at the headline shape (GPU needed).

Measures the step time of perturbations and of topology moves, records a trace of the standard
ladder, and replays it under the model "K workgroups evaluate proposals s .. s+K-1 against the
committed state; the round ends with the first accepted one"."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from procgen_amd import native as nt            # noqa: E402
from procgen_amd import workloads as wlm        # noqa: E402
from procgen_amd.engine import GraphEngine      # noqa: E402


def timed(wl, n, R, mode, iters=400):
    eng = GraphEngine(n, wl["ecap"], R, 1, seed="spec", mutation_mode=mode)
    eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
    eng.run(1); eng.run(200); eng.sync(); eng.kernel_time(reset=True)
    eng.run(iters); eng.sync()
    ms, launches = eng.kernel_time()
    st = eng.stats()
    eng.close()
    return ms / iters * 1e3, st["accepted"] / st["proposals"]


def main():
    n, R = 10000, 64
    wl = wlm.graph_workload("c2", n, R, 1)
    tp, acc_p = timed(wl, n, R, nt.MUT_PERTURB_ONLY)
    hot = dict(wl); hot["temperatures"] = np.full_like(wl["temperatures"], 5.0)
    th, acc_h = timed(hot, n, R, nt.MUT_ADAPTIVE)
    cold = dict(wl); cold["temperatures"] = np.full_like(wl["temperatures"], 0.5)
    tc, acc_c = timed(cold, n, R, nt.MUT_ADAPTIVE)
    tt = (th - 0.2 * tp) / 0.8
    print(f"us/step: perturb-only {tp:.2f} (acc {acc_p:.2f}), all-hot {th:.2f} (acc {acc_h:.2f}), all-cold {tc:.2f} (acc {acc_c:.2f}) -> topology move ~{tt:.2f}")
    iters = 601
    eng = GraphEngine(n, wl["ecap"], R, 1, seed="procgen-bench/c2")
    eng.upload(wl["pos"], wl["edges"], wl["ecount"], wl["temperatures"], wl["kind"], wl["value"])
    eng.enable_trace(iters, 8)
    eng.run(iters); eng.sync()
    tr, _ = eng.read_trace(iters, 0)
    eng.close()
    tr = tr[1:]                                    # [600, 64] slot order
    acc = (tr["flags"] & 4) != 0
    kind = tr["kind"]
    print("accept rate by slot (0 = hottest):", np.round(acc.mean(0), 2).tolist())
    print("perturb share by slot:", np.round((kind == 0).mean(0), 2).tolist())
    for k in (0, 1, 2):
        m = kind == k
        print(f"kind {k}: share {m.mean():.2f}, accept {acc[m].mean():.2f}")
    cost = np.where(kind == 0, tp, tt)
    base = cost.sum(0)
    print(f"model, 1 workgroup: slowest slot {base.max():.0f} us per 600 steps, mean {base.mean():.0f}")
    for K in (2, 3, 4, 5, 6):
        for comm in (1.0, 2.0):
            tot = np.zeros(R)
            for s in range(R):
                i = 0
                while i < 600:
                    j = i
                    while j < min(i + K, 600) - 1 and not acc[j, s]:
                        j += 1
                    tot[s] += cost[i:j + 1, s].max() + comm
                    i = j + 1
            print(f"K={K} comm={comm}: slowest slot {tot.max():.0f} us, mean {tot.mean():.0f}, "
                  f"speedup of the slowest {base.max() / tot.max():.2f}; by slot/8: {np.round(tot.reshape(8, 8).mean(1)).tolist()}")


if __name__ == "__main__":
    main()

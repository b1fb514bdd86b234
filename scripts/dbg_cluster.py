import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import parity_util as pu
from procgen_amd import workloads as wlm
n,T,G,iters,f=300,6,2,300,34
wl = wlm.graph_workload("trace", n, T, G, layout="box", index=n)
a = pu.gpu_run(wl, iters, f, seed=f"trace-{n}", chunks=[100,200], cluster_size=1)
for C in (2,3,4):
    b = pu.gpu_run(wl, iters, f, seed=f"trace-{n}", chunks=[100,200], cluster_size=C)
    print("C",C,"status",b["stats"]["status_bits"])
    first=None
    for fld in pu.TRACE_FIELDS:
        d = np.argwhere(a["trace"][fld]!=b["trace"][fld])
        if len(d): 
            print(fld, "first diff", d[0], "count", len(d))
            if first is None or tuple(d[0])<first: first=tuple(d[0])
    if first:
        s,c=first
        for ss in range(max(0,s-2), s+2):
            print(ss, c, "C1", a["trace"][ss,c], "\n      C%d"%C, b["trace"][ss,c])
    sw = np.argwhere(a["swap_trace"]!=b["swap_trace"])
    print("swap diffs", sw[:3])

#!/bin/bash
# usage: scripts/pmc_any.sh <outname> "<counters>" <python script and args...>   (on the GPU box, repo root)
# One rocprofv3 --pmc pass over any script that drives the graph engine; prints per-dispatch sums of the
# largest dispatch of a step kernel.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
name=$1; counters=$2; shift 2
cd /tmp
timeout -s KILL 150 rocprofv3 --pmc $counters --output-format csv -d $R/gpurun_out/$name -- python3 "$@" > $R/gpurun_out/$name.log 2>&1
python3 - "$R/gpurun_out/$name" <<'PY'
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if any(k in r["Kernel_Name"] for k in ("steps_kernel", "delta_kernel", "cluster_kernel")):
        agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
best = max(agg, key=lambda d: max(agg[d].values()))
print("dispatch", best, {k: f"{v:.4g}" for k, v in agg[best].items()})
PY
tail -1 $R/gpurun_out/$name.log

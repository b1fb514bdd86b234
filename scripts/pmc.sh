#!/bin/bash
# usage: scripts/pmc.sh <outname> <counters...>   (run on the GPU box from the repo root)
# One rocprofv3 --pmc pass over a short bench run; per-dispatch sums of the step kernel are printed.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
name=$1; shift
cd /tmp
timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/$name -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $R/gpurun_out/$name.log 2>&1
python3 - "$R/gpurun_out/$name" <<'PY'
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if any(k in r["Kernel_Name"] for k in ("steps_kernel", "delta_kernel", "cluster_kernel")):
        agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
last = sorted(agg, key=int)[-1]
print("dispatch", last, {k: f"{v:.4g}" for k, v in agg[last].items()})
import json
json.dump({"dispatch": int(last), "counters": dict(agg[last]), "note": "sums over the last step-kernel dispatch of bench.py --steps 2 (one launch = 12 800 proposals at the headline shape)"}, open(sys.argv[1] + ".json", "w"), indent=1)
PY
grep -o '"kernel_ms_avg": [0-9.]*' $R/gpurun_out/$name.log

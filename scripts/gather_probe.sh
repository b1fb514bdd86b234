#!/bin/bash
# usage: scripts/gather_probe.sh   (on the GPU box, from the repo root; scripts/gather_probe built beforehand)
# The probe alone (rates), then one rocprofv3 --pmc FETCH_SIZE pass over it (counter bytes per chunk).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 120 $R/scripts/gather_probe 8 > $R/gpurun_out/gather_probe.txt 2>&1
cd /tmp
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/gather_pmc -- $R/scripts/gather_probe 8 > $R/gpurun_out/gather_pmc.log 2>&1
python3 - "$R" <<'PY' >> $R/gpurun_out/gather_probe.txt
import csv, collections, glob, sys
R = sys.argv[1]
f = glob.glob(f"{R}/gpurun_out/gather_pmc/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        agg[(int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0])] += float(r["Counter_Value"])
print("rocprofv3 --pmc FETCH_SIZE per dispatch (KB of 1024 B as reported; the timed dispatch of each size gathers 256 * 16 * 256 / W * 256 chunks):")
for (d, k), v in sorted(agg.items()):
    print(f"  dispatch {d} {k}: FETCH_SIZE {v:.0f} KB")
PY
cat $R/gpurun_out/gather_probe.txt

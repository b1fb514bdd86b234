#!/bin/bash
# usage: scripts/pmc_traffic.sh   (run on the GPU box from the repo root)
#        BENCH_ARGS="--workload chunks" OUT=pmc_traffic_chunks scripts/pmc_traffic.sh   for another workload
# Two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass) over a short bench run;
# writes gpurun_out/pmc_traffic.json with the HBM bytes per launch of the step kernel.
# FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1 KB as rocprofv3 reports them; FETCH_SIZE is doubled
# (gfx950 tallies 128-byte read requests at 64 bytes).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${OUT:-pmc_traffic}
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 150 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/${OUT}_$c -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline $BENCH_ARGS > $R/gpurun_out/${OUT}_$c.log 2>&1
done
python3 - "$R" "$OUT" <<'PY'
import csv, collections, glob, json, sys
R = sys.argv[1]
OUT = sys.argv[2]
out = {}
kernel = None
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{R}/gpurun_out/{OUT}_{c}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("steps_kernel", "delta_kernel", "cluster_kernel", "big_kernel", "bigc_kernel")) and r["Counter_Name"] == c:
            agg[r["Dispatch_Id"]] += float(r["Counter_Value"])
            kernel = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").strip()
    last = sorted(agg, key=int)[-1]
    out[c] = agg[last]
ppl = None
for line in open(f"{R}/gpurun_out/{OUT}_WRITE_SIZE.log"):
    if line.startswith("{") and "proposals_per_launch" in line:
        ppl = json.loads(line)["roofline"]["proposals_per_launch"]
rec = {"kernel": kernel, "fetch_size_kb": out["FETCH_SIZE"], "write_size_kb": out["WRITE_SIZE"], "proposals_per_launch": ppl,
       "hbm_bytes_per_launch": 2.0 * out["FETCH_SIZE"] * 1024.0 + out["WRITE_SIZE"] * 1024.0,
       "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes, last launch of bench.py --steps 2; "
               "bytes = 2 x FETCH_SIZE (gfx950 read-request correction) + WRITE_SIZE, KB = 1024 B"}
json.dump(rec, open(f"{R}/gpurun_out/{OUT}.json", "w"), indent=1)
print(rec)
PY

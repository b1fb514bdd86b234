#!/bin/bash
# usage: scripts/round_profiles.sh <tag>   (on the GPU box, from the repo root; e.g. tag = r02_final)
# Everything the profiles/ directory holds for one round: kernel-trace statistics of the default bench run, the
# bench lines of the three measured shapes, HBM traffic of each (two --pmc passes) and the SQ counters of the headline.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=$1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -- python3 $R/bench.py > $R/gpurun_out/${T}_bench.json 2> $R/gpurun_out/${T}_bench.err
cp $(ls $R/gpurun_out/${T}_prof/*/*kernel_stats.csv | head -1) $R/gpurun_out/${T}_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof_chunks -- python3 $R/bench.py --workload chunks --no-cpu-baseline > /dev/null 2> $R/gpurun_out/${T}_prof_chunks.err
cp $(ls $R/gpurun_out/${T}_prof_chunks/*/*kernel_stats.csv | head -1) $R/gpurun_out/${T}_chunks_kernel_stats.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof_1m -- python3 $R/bench.py --nodes 1000000 --replicas 16 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/${T}_prof_1m.err
cp $(ls $R/gpurun_out/${T}_prof_1m/*/*kernel_stats.csv | head -1) $R/gpurun_out/${T}_1m_kernel_stats.csv
cd $R
# the round-end driver's own command line (20 timed intervals after 5)
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver_line.json 2> gpurun_out/${T}_bench_driver_line.err
timeout -k 10 300 python3 bench.py --workload chunks > gpurun_out/${T}_bench_chunks.json 2> gpurun_out/${T}_bench_chunks.err
timeout -k 10 400 python3 bench.py --nodes 1000000 --replicas 16 > gpurun_out/${T}_bench_config3_shape.json 2> gpurun_out/${T}_bench_config3.err
timeout -k 10 400 python3 bench.py --nodes 1000 --replicas 16 --ladders 512 > gpurun_out/${T}_bench_config4_shape.json 2> gpurun_out/${T}_bench_config4.err
bash scripts/pmc_traffic.sh > gpurun_out/${T}_pmc_traffic.txt 2>&1
BENCH_ARGS="--workload chunks" OUT=pmc_traffic_chunks bash scripts/pmc_traffic.sh > gpurun_out/${T}_pmc_traffic_chunks.txt 2>&1
BENCH_ARGS="--nodes 1000000 --replicas 16" OUT=pmc_traffic_1m bash scripts/pmc_traffic.sh > gpurun_out/${T}_pmc_traffic_1m.txt 2>&1
bash scripts/pmc.sh pmc_sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT > gpurun_out/${T}_pmc_sq.txt 2>&1
echo done

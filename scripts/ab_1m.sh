#!/bin/bash
# times library builds (procgen_amd/lib_<name>; "lib" = the shipped one) on the million-node shape (16 replicas x 10^6
# nodes, one swap interval per launch) with bench.py's own line, the CPU baseline left out.  usage: ab_1m.sh name...
for v in "$@"; do
  p=procgen_amd/lib_$v/libgpu_compute.so; [ "$v" = lib ] && p=procgen_amd/lib/libgpu_compute.so
  PROCGEN_PT_LIB=$p timeout -k 10 280 python bench.py --nodes 1000000 --replicas 16 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ' | sed "s/^/$v /" || exit 1
  echo
done

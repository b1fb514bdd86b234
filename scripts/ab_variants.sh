#!/bin/bash
# times the library builds named on the command line (procgen_amd/lib_<name>) against the shipped one, twice each
for v in "$@" lib; do
  p=procgen_amd/lib_$v/libgpu_compute.so; [ "$v" = lib ] && p=procgen_amd/lib/libgpu_compute.so
  for rep in 1 2; do
    PROCGEN_PT_LIB=$p timeout -k 10 100 python scripts/step_time.py ${AB_ARGS:-10000 64 800 0} 2>/dev/null | grep -o "us/step=[0-9.]*" | sed "s/^/$v /" || exit 1
  done
done

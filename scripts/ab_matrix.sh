#!/bin/bash
# times library builds (procgen_amd/lib_<name>; "lib" = the shipped one) on three proposal mixes of the headline shape:
# the bench ladder, perturbations only, and an all-hot ladder (four topology moves in five).  usage: ab_matrix.sh name...
for v in "$@"; do
  p=procgen_amd/lib_$v/libgpu_compute.so; [ "$v" = lib ] && p=procgen_amd/lib/libgpu_compute.so
  for mix in "10000 64 800 0" "10000 64 800 1" "10000 64 800 0 0 4.9 5.0"; do
    PROCGEN_PT_LIB=$p timeout -k 10 100 python scripts/step_time.py $mix 2>/dev/null | grep -o "perturb_only=[01] T=[^ ]*\|us/step=[0-9.]*\|accept=[0-9.]*" | tr '\n' ' ' | sed "s/^/$v /" || exit 1
    echo
  done
done

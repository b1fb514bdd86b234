#!/bin/bash
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in lib lib_w2; do
  export PROCGEN_PT_LIB=$R/procgen_amd/$v/libgpu_compute.so
  cd $R && bash scripts/pmc.sh pmc_ab_$v SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
done

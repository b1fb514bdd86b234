#!/bin/bash
# The bounds of the subtree walk (PROCGEN_PT_DELTA_WALK=members,entering,rounds) against the step time of the headline
# shape, bench ladder and all-hot ladder, with the per-kind ticks of a -DPT_STAMPS_LIGHT build.  usage: walk_sweep.sh
for w in "48,128,1024" "48,128,12" "96,128,16" "256,128,16" "256,128,24" "256,128,40" "256,128,1024"; do
  echo "== walk $w"
  for mix in "10000 64 800 0" "10000 64 800 0 0 4.9 5.0"; do
    PROCGEN_PT_DELTA_WALK=$w timeout -k 10 100 python scripts/step_time.py $mix 2>/dev/null | grep -o "T=[^ ]*\|us/step=[0-9.]*" | tr '\n' ' ' || exit 1
    echo
  done
  PROCGEN_PT_DELTA_WALK=$w timeout -k 10 100 python scripts/kind_profile.py 10000 64 2000 2>/dev/null | grep "search\|on average" || exit 1
done

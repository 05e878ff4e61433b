#!/bin/bash
# A/B of two builds of the library inside ONE gpurun call (boxes differ by a few per cent): the shipped one against lib/libvo_mi355x_b.so
# (csrc/Makefile: make OBJDIR=build_b OUT=../lib/libvo_mi355x_b.so EXTRA=...), interleaved.  usage: tools/ab_lib.sh ["bench args"] [rounds]
ARGS=${1:---no-extras --no-cpu-baseline --regions 3 --steps 50}
N=${2:-2}
B=$PWD/visual-odom-pipeline_amd/lib/libvo_mi355x_b.so
for r in $(seq $N); do
  order="a b"; [ $((r % 2)) = 0 ] && order="b a"          # a b b a a b ...: a box warms up over the first runs
  for which in $order; do
    if [ $which = b ]; then export VO_MI355X_LIB=$B; else unset VO_MI355X_LIB; fi
    out=$(timeout 300 python3 bench.py $ARGS --full-line --extras-file /dev/null 2>/dev/null | tail -1)
    python3 - "$out" $which <<'P'
import json, sys
try:
    d = json.loads(sys.argv[1]); print("lib %s: %9.1f frames/s  %.4f ms/step  ba stage %.4f ms" % (sys.argv[2], d["value"], d["ms_per_step"], (d.get("stage_ms_per_batched_launch_group") or {}).get("ba", float("nan"))))
except Exception as e:
    print("lib %s: failed (%s) %s" % (sys.argv[2], e, sys.argv[1][:200]))
P
  done
done
unset VO_MI355X_LIB

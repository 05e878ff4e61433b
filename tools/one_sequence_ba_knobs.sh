#!/bin/bash
# Small batches: the wave-private BA kernels (VO_BA_V2=1) against the lane-per-observation ones (VO_BA_V2=0)
run() {
  out=$(env "$@" timeout 150 python bench.py --no-extras --no-cpu-baseline $EXTRA 2>/dev/null | tail -1)
  python - "$out" "$* $EXTRA" <<'P'
import json, sys
d = json.loads(sys.argv[1]); print("%-70s %9.1f frames/s  %.4f ms/step" % (sys.argv[2], d["value"], d["ms_per_step"]))
P
}
for b in 1 2 3 4 8; do
  EXTRA="--seqs $b --steps 300 --ba-iters 10"
  run VO_BA_V2=1
  run VO_BA_V2=0
done
for b in 1 2 4; do
  EXTRA="--workload pipeline --seqs $b --steps 100"
  run VO_BA_V2=1
  run VO_BA_V2=0
done

#!/bin/bash
# Small batches: the wave-private BA kernels (vo_tuning.ba_kernels = 2) against the lane-per-observation ones (ba_kernels = 1)
run() {
  out=$(timeout 150 python bench.py --no-extras --no-cpu-baseline --full-line $EXTRA --tune $1 2>/dev/null | tail -1)
  python - "$out" "$* $EXTRA" <<'P'
import json, sys
d = json.loads(sys.argv[1]); print("%-70s %9.1f frames/s  %.4f ms/step" % (sys.argv[2], d["value"], d["ms_per_step"]))
P
}
for b in 1 2 3 4 8; do
  EXTRA="--seqs $b --steps 300 --ba-iters 10"
  run ba_kernels=2
  run ba_kernels=1
done
for b in 1 2 4; do
  EXTRA="--workload pipeline --seqs $b --steps 100"
  run ba_kernels=2
  run ba_kernels=1
done

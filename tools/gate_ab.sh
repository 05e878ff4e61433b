#!/bin/bash
# A/B: the next frame's tracker launch gated behind the first K LM groups of the previous frame's chain (vo_tuning.gate_groups), stream A leaving R
# compute units free (vo_tuning.reserve_cus).  usage: tools/gate_ab.sh
run() {
  out=$(timeout 150 python bench.py --no-extras --no-cpu-baseline --full-line --steps 60 $EXTRA ${1:+--tune $1} 2>/dev/null | tail -1)
  python - "$out" "$* $EXTRA" <<'P'
import json, sys
try:
    d = json.loads(sys.argv[1]); print("%-66s %9.1f frames/s  %.4f ms/step  capped %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["config"]["ba_solves_stopped_by_lm_max_iters"]))
except Exception as e:
    print("%-66s failed: %s" % (sys.argv[2], sys.argv[1][:200]))
P
}
for seqs in 1 8 16 32; do
  EXTRA="--ba-iters 30 --seqs $seqs --steps 200"
  run ""
  run gate_groups=5,reserve_cus=32
  run gate_groups=5,reserve_cus=-1
done

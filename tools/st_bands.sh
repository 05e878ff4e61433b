#!/bin/bash
# k_st_eig_fused band height (vo_tuning.st_band_rows; 0 = the rule) in the default command: frames/s and the re-detection's launch group
for rb in 0 376 188 126 94 76 63 0; do
  out=$(timeout 120 python bench.py --no-extras --no-cpu-baseline --full-line --steps 60 --tune st_band_rows=$rb 2>/dev/null | tail -1)
  python - "$out" $rb <<'P'
import json, sys
d = json.loads(sys.argv[1]); print("st_band_rows=%s %9.1f frames/s  %.4f ms/step  st %.4f ba %.4f" % (sys.argv[2], d["value"], d["ms_per_step"], d["stage_ms_per_batched_launch_group"]["shi_tomasi"], d["stage_ms_per_batched_launch_group"]["ba"]))
P
done

# the library's rules against their neighbours on the current build (vo_tuning through bench.py --tune): one line per setting, the rule first and last
run() {
  out=$(timeout 200 python3 bench.py --no-extras --no-cpu-baseline --full-line --extras-file /dev/null --steps 60 --regions 5 ${1:+--tune $1} 2>/dev/null | tail -1)
  python3 - "$out" "$1" <<'P'
import json, sys
try:
    d = json.loads(sys.argv[1]); print("%-28s %9.1f frames/s  %.4f ms/step" % (sys.argv[2] or "rule", d["value"], d["ms_per_step"]))
except Exception as e:
    print("%-28s failed: %s" % (sys.argv[2], sys.argv[1][:120]))
P
}
run ""
for t in klt_waves=5 klt_waves=4 xcd_remap_off=1 st_band_rows=126 st_band_rows=76 st_two_kernels=1 ba_workgroups=1 ba_workgroups=4 ba_lanes=8 ba_fold=1; do run $t; done
run ""

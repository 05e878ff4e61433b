run() {
  out=$(timeout 200 python3 bench.py --no-extras --no-cpu-baseline --full-line --extras-file /dev/null --steps 60 --regions 5 ${1:+--tune $1} 2>/dev/null | tail -1)
  python3 - "$out" "$1" <<'P'
import json, sys
try:
    d = json.loads(sys.argv[1]); print("%-40s %9.1f frames/s  %.4f ms/step  layout %s" % (sys.argv[2] or "rule", d["value"], d["ms_per_step"], d["config"]["stream_layout"]))
except Exception as e:
    print("%-40s failed: %s" % (sys.argv[2], sys.argv[1][:200]))
P
}
run ""
run gate_groups=3
run gate_groups=5
run gate_groups=6
run ""
run gate_groups=5
run gate_groups=4,reserve_cus=64
run gate_groups=3
run ""

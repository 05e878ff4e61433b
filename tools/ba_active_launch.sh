#!/bin/bash
# Time of a FULLY ACTIVE BA launch group (every problem of the batch in its first LM iterations): rocprofv3 over tools/phase_probe_batch.py
#   tools/ba_active_launch.sh [batch] [tag]
B=${1:-256}; TAG=${2:-act}
OUT=$PWD/gpurun_out; SCRIPT=$PWD/tools/phase_probe_batch.py
export TMPDIR=/tmp
cd /tmp; rm -rf $OUT/${TAG}_act
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_act -o ks -- python3 $SCRIPT $B > $OUT/${TAG}_act.log 2>&1
cd - >/dev/null
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/${TAG}_act/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
by = {}
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
    by.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in by.items():
    if n.startswith("k_ba"): print("%-28s n %3d  first4 %s  min %.1f max %.1f" % (n, len(v), " ".join("%.1f" % x for x in v[:6]), min(v), max(v)))
PY
rm -rf $OUT/${TAG}_act

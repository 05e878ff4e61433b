#!/bin/bash
# Per-kernel time of one batched context without cross-stream overlap (rocprofv3 --kernel-trace --stats):
#   tools/kstats.sh [tag]      -> gpurun_out/<tag>_kstats.txt
TAG=${1:-k}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
BENCH="$PWD/bench.py"
cd /tmp
rm -rf $OUT/${TAG}_ks
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks -o ks -- python3 $BENCH --steps 30 --warmup 5 --regions 1 --no-extras --seqs 32 --ctxs 1 --host-threads 1 --side-stream off --no-cpu-baseline > $OUT/${TAG}_ks.log 2>&1
cd - > /dev/null
python3 - <<PY > $OUT/${TAG}_kstats.txt
import csv, glob
f = sorted(glob.glob("$OUT/${TAG}_ks/**/*kernel_stats.csv", recursive=True))[-1]
tot = 0
rows = list(csv.DictReader(open(f)))
steps = max(1, sum(int(r["Calls"]) for r in rows if "k_klt_track" in r["Name"]))     # one KLT launch per step (warm-up, timed and untimed steps alike)
for r in rows:
    n = r["Name"].split("(")[0].replace("void ", "")[:24]
    t = float(r["TotalDurationNs"]) / 1e3
    tot += t
    print("%-24s calls %5s avg %8.1f us  per-step %7.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, t / steps))
print("sum per 32-frame step: %.1f us over %d steps (adaptive-budget, fixed-budget and stationary-BA regions mixed)" % (tot / steps, steps))
PY
cat $OUT/${TAG}_kstats.txt

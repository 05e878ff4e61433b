#!/bin/bash
# tools/kstats.sh for the whole Pipeline.step on the device (--workload pipeline: track table, RANSAC-P3P pose, DLT, BA, re-detection)
TAG=${1:-kp}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp VO_SIDE_STREAM=0
BENCH="$PWD/bench.py"
cd /tmp
rm -rf $OUT/${TAG}_ks
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks -o ks -- python3 $BENCH --steps 30 --warmup 5 --regions 1 --no-extras --seqs 32 --ctxs 1 --host-threads 1 --side-stream off --no-cpu-baseline --workload pipeline > $OUT/${TAG}_ks.log 2>&1
cd - > /dev/null
python3 - <<PY > $OUT/${TAG}_kstats.txt
import csv, glob
f = sorted(glob.glob("$OUT/${TAG}_ks/**/*kernel_stats.csv", recursive=True))[-1]
tot = 0
rows = list(csv.DictReader(open(f)))
steps = 35 + 20 + 1
for r in rows:
    n = r["Name"].split("(")[0].replace("void ", "")[:26]
    t = float(r["TotalDurationNs"]) / 1e3
    tot += t
    print("%-26s calls %5s avg %8.1f us  per-step %7.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, t / steps))
print("sum per 32-frame step: %.1f us" % (tot / steps))
PY
cat $OUT/${TAG}_kstats.txt

#!/bin/bash
# Per-kernel time of one batched context without cross-stream overlap (rocprofv3 --kernel-trace --stats):
#   tools/kstats.sh [tag]      -> gpurun_out/<tag>_kstats.txt
TAG=${1:-k}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp VO_SIDE_STREAM=0
BENCH="$PWD/bench.py"
cd /tmp
rm -rf $OUT/${TAG}_ks
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks -o ks -- python3 $BENCH --steps 30 --warmup 5 --regions 1 --no-extras --seqs 32 --ctxs 1 --host-threads 1 --side-stream off --no-cpu-baseline > $OUT/${TAG}_ks.log 2>&1
cd - > /dev/null
python3 - <<PY > $OUT/${TAG}_kstats.txt
import csv, glob
f = sorted(glob.glob("$OUT/${TAG}_ks/**/*kernel_stats.csv", recursive=True))[-1]
tot = 0
steps = 35 + 20 + 1   # warm-up + timed + the 20 untimed stage-breakdown steps + the first push
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("void ", "")[:24]
    t = float(r["TotalDurationNs"]) / 1e3
    tot += t
    print("%-24s calls %5s avg %8.1f us  per-step %7.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, t / steps))
print("sum per 32-frame step: %.1f us" % (tot / steps))
PY
cat $OUT/${TAG}_kstats.txt

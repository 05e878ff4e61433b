#!/bin/bash
# Per-kernel time of the default batch (256 sequences, one context) without cross-stream overlap (rocprofv3 --kernel-trace --stats):
#   tools/kstats256.sh [tag]      -> gpurun_out/<tag>_kstats256.txt
TAG=${1:-k}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
BENCH="$PWD/bench.py"
cd /tmp
rm -rf $OUT/${TAG}_ks256
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks256 -o ks -- python3 $BENCH --steps 20 --warmup 5 --regions 1 --no-extras --seqs 256 --ctxs 1 --host-threads 1 --side-stream off --no-cpu-baseline > $OUT/${TAG}_ks256.log 2>&1
cd - > /dev/null
python3 - <<PY > $OUT/${TAG}_kstats256.txt
import csv, glob
f = sorted(glob.glob("$OUT/${TAG}_ks256/**/*kernel_stats.csv", recursive=True))[-1]
tot = 0
rows = list(csv.DictReader(open(f)))
steps = max(1, sum(int(r["Calls"]) for r in rows if "k_klt_track" in r["Name"]))
for r in rows:
    n = r["Name"].split("(")[0].replace("void ", "")[:24]
    t = float(r["TotalDurationNs"]) / 1e3
    tot += t
    print("%-24s calls %5s avg %8.1f us  per-step %7.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, t / steps))
print("sum per 256-frame step: %.1f us over %d steps" % (tot / steps, steps))
PY
rm -rf $OUT/${TAG}_ks256/*/*kernel_trace.csv $OUT/${TAG}_ks256/*kernel_trace.csv
cat $OUT/${TAG}_kstats256.txt

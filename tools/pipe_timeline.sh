#!/bin/bash
# GPU timeline of ONE closed-loop step at a batch (kernel trace of bench.py --workload pipeline): per kernel name, first start / last end relative to
# the tracker launch of the step, summed busy time.   tools/pipe_timeline.sh [seqs] [window]
SEQS=${1:-256}; WIN=${2:-4}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
BENCH="$PWD/bench.py"
cd /tmp; rm -rf $OUT/ptl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/ptl -o ks -- python3 $BENCH --workload pipeline --seqs $SEQS --pipe-window $WIN --steps 12 --warmup 6 --regions 1 --no-cpu-baseline > $OUT/ptl.log 2>&1
cd - > /dev/null
python3 - <<PY | tee $OUT/pipe_timeline_${SEQS}_w${WIN}.txt
import csv, glob, collections
f = sorted(glob.glob("$OUT/ptl/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
klt = [i for i, r in enumerate(rows) if "k_klt_track" in r["Kernel_Name"]]
a, b = klt[-4], klt[-3]            # one steady-state step: from one tracker launch to the next
t0 = int(rows[a]["Start_Timestamp"])
agg = collections.OrderedDict()
for r in rows[a:b]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:30]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    x = agg.setdefault(n, [s, e, 0.0, 0])
    x[0] = min(x[0], s); x[1] = max(x[1], e); x[2] += e - s; x[3] += 1
print("step = %.1f us (tracker launch to tracker launch), %d sequences, window %d" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, $SEQS, $WIN))
for n, (s, e, busy, cnt) in agg.items():
    print("%-30s launches %3d  first start %8.1f  last end %8.1f  summed duration %8.1f us" % (n, cnt, s, e, busy))
PY
rm -rf $OUT/ptl

#!/bin/bash
# Per-kernel time of the closed-loop Pipeline.step on the device (bench.py --workload pipeline), one batched context of 32 sequences
# so that no two kernels overlap (rocprofv3 --kernel-trace --stats):
#   tools/kstats_pipeline.sh [tag] [extra bench flags]      -> gpurun_out/<tag>_kstats.txt
TAG=${1:-kp}
shift
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
BENCH="$PWD/bench.py"
cd /tmp
rm -rf $OUT/${TAG}_ks
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks -o ks -- python3 $BENCH --steps 40 --warmup 10 --regions 1 --seqs 32 --ctxs 1 --host-threads 1 --workload pipeline --no-cpu-baseline --full-line --extras-file /dev/null "$@" > $OUT/${TAG}_ks.log 2>&1
cd - > /dev/null
python3 - <<PY > $OUT/${TAG}_kstats.txt
import csv, glob, json
f = sorted(glob.glob("$OUT/${TAG}_ks/**/*kernel_stats.csv", recursive=True))[-1]
tot = 0
rows = list(csv.DictReader(open(f)))
steps = max(1, sum(int(r["Calls"]) for r in rows if "k_klt_track" in r["Name"]))
for r in rows:
    n = r["Name"].split("(")[0].replace("void ", "")[:26]
    t = float(r["TotalDurationNs"]) / 1e3
    tot += t
    print("%-26s calls %5s avg %8.1f us  per-step %7.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, t / steps))
print("sum per step of the batch: %.1f us  (incl. the one-off bootstrap launches)" % (tot / steps))
try:
    d = json.loads([l for l in open("$OUT/${TAG}_ks.log") if l.startswith("{")][-1])["pipeline"]
    print("bench line of this run:", {k: d[k] for k in ("frames_per_s", "ms_per_step", "ba_window", "resurrection_of_dead_landmarks", "mean_tracked_keypoints", "mean_landmark_entries", "mean_ba_observations", "ba_iterations_histogram")})
except Exception as e:
    print("no bench line:", e)
PY
cat $OUT/${TAG}_kstats.txt

TAG=kp1
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
BENCH="$PWD/bench.py"
cd /tmp
rm -rf $OUT/${TAG}_ks
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks -o ks -- python3 $BENCH --steps 100 --warmup 10 --regions 1 --seqs 1 --ctxs 1 --host-threads 1 --workload pipeline --no-cpu-baseline --full-line --extras-file /dev/null "$@" > $OUT/${TAG}_ks.log 2>&1
cd - > /dev/null
python3 - <<PY
import csv, glob, json
f = sorted(glob.glob("$OUT/${TAG}_ks/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
steps = max(1, sum(int(r["Calls"]) for r in rows if "k_klt_track" in r["Name"]))
tot=0
for r in rows:
    n = r["Name"].split("(")[0].replace("void ", "")[:26]
    t = float(r["TotalDurationNs"]) / 1e3; tot+=t
    print("%-26s calls %5s avg %8.1f us  per-step %7.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, t / steps))
print("sum per frame: %.1f us" % (tot/steps))
d = json.loads([l for l in open("$OUT/${TAG}_ks.log") if l.startswith("{")][-1])["pipeline"]
print({k: d[k] for k in ("frames_per_s", "ms_per_step", "ba_iterations_histogram")})
PY

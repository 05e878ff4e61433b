#!/bin/bash
# Per launch-group index of a step (1..budget): mean duration of the BA kernels at the default batch, one stream (no overlap)
#   tools/ba_groups.sh [tag] [seqs]
TAG=${1:-g}; SEQS=${2:-256}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
BENCH="$PWD/bench.py"
cd /tmp; rm -rf $OUT/${TAG}_grp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_grp -o ks -- python3 $BENCH --steps 20 --warmup 5 --regions 1 --no-extras --seqs $SEQS --ctxs 1 --host-threads 1 --side-stream off --no-cpu-baseline > $OUT/${TAG}_grp.log 2>&1
cd - > /dev/null
python3 - <<PY | tee $OUT/${TAG}_groups.txt
import csv, glob, collections
f = sorted(glob.glob("$OUT/${TAG}_grp/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
grp = collections.defaultdict(lambda: collections.defaultdict(list))
idx = collections.Counter()
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    if n == "k_ba_finalize": idx.clear(); continue
    if n.startswith("k_ba_"):
        idx[n] += 1
        grp[n][idx[n]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n in grp:
    print("%-16s" % n, " ".join("%d:%.0f" % (g, sum(v) / len(v)) for g, v in sorted(grp[n].items())), " (us, mean over %d steps)" % len(grp[n][1]))
    print("%-16s" % "  median", " ".join("%d:%.0f" % (g, sorted(v)[len(v) // 2]) for g, v in sorted(grp[n].items())))
    print("%-16s" % "  max", " ".join("%d:%.0f" % (g, max(v)) for g, v in sorted(grp[n].items())))
PY
rm -rf $OUT/${TAG}_grp

#!/bin/bash
# Per-launch duration of the pyramid kernels (k_pad_level0, k_scharr_pyrdown per level) of a 256-sequence step on one stream:
#   tools/pyramid_levels.sh [tag]   -> gpurun_out/<tag>_pyramid_levels.txt
TAG=${1:-k}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
BENCH="$PWD/bench.py"
cd /tmp
rm -rf $OUT/${TAG}_pyr
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_pyr -o ks -- python3 $BENCH --steps 10 --warmup 3 --regions 1 --no-extras --seqs 256 --ctxs 1 --host-threads 1 --side-stream off --no-cpu-baseline > $OUT/${TAG}_pyr.log 2>&1
cd - > /dev/null
python3 - <<PY > $OUT/${TAG}_pyramid_levels.txt
import csv, glob, collections
f = sorted(glob.glob("$OUT/${TAG}_pyr/**/*kernel_trace.csv", recursive=True))[-1]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "scharr" in n or "pad_level0" in n or "st_eig" in n or "st_discs" in n:
        d[(n.split("(")[0].replace("void ", "")[:28], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda kv: -kv[0][1]):
    v = v[len(v) // 3:]
    print("%-28s workgroups.x %7d  launches %3d  avg %8.1f us  min %8.1f" % (k[0], k[1], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3))
PY
rm -rf $OUT/${TAG}_pyr
cat $OUT/${TAG}_pyramid_levels.txt

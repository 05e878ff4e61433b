#!/bin/bash
# VALU instruction counters of the two KLT kernels on the same launch (32 sequences x 2000 keypoints): one wave per keypoint
# (k_klt_track<6>, shipped) against two keypoints per wave (k_klt_track2<3>, vo_tuning.klt_pair = 3: a library built with -DVO_EXPERIMENTS)  -> gpurun_out/klt_pair_counters.txt
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
BENCH="$PWD/bench.py"
cd /tmp
for MODE in 0 3; do
  TUNE=""; [ "$MODE" != 0 ] && TUNE="--tune klt_pair=$MODE"      # (needs a library built with -DVO_EXPERIMENTS)
  rm -rf $OUT/kpair_$MODE
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/kpair_$MODE -o c -- python3 $BENCH --steps 4 --warmup 2 --regions 1 --no-extras --no-cpu-baseline --seqs 32 --ctxs 1 $TUNE > $OUT/kpair_$MODE.log 2>&1
  rm -rf $OUT/kpair_t$MODE
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kpair_t$MODE -o t -- python3 $BENCH --steps 20 --warmup 5 --regions 1 --no-extras --no-cpu-baseline --seqs 32 --ctxs 1 --side-stream off $TUNE > $OUT/kpair_t$MODE.log 2>&1
done
cd - > /dev/null
python3 - <<PY > $OUT/klt_pair_counters.txt
import csv, glob
from collections import defaultdict
print("k_klt_track (one wave per keypoint) vs k_klt_track2 (two keypoints per wave), one launch = 32 sequences x 2000 keypoints, bit-identical outputs")
for mode in (0, 3):
    f = sorted(glob.glob("$OUT/kpair_%d/**/*counter_collection.csv" % mode, recursive=True))[-1]
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "k_klt_track" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    m = {k: v[0] / max(v[1], 1) for k, v in acc.items()}
    t = sorted(glob.glob("$OUT/kpair_t%d/**/*kernel_stats.csv" % mode, recursive=True))[-1]
    us = [float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(t)) if "k_klt_track" in r["Name"]]
    print("klt_pair=%d: SQ_INSTS_VALU %.4g per launch = %.0f per keypoint, SQ_WAVES %.0f, SQ_BUSY_CYCLES %.4g, GRBM_GUI_ACTIVE %.4g, %.1f us per launch (rocprofv3 --stats, single stream)"
          % (mode, m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_VALU", 0) / 64000.0, m.get("SQ_WAVES", 0), m.get("SQ_BUSY_CYCLES", 0), m.get("GRBM_GUI_ACTIVE", 0), us[0] if us else -1))
PY
cat $OUT/klt_pair_counters.txt

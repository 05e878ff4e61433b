#!/bin/bash
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
BENCH="$PWD/bench.py"
ARGS="--steps 4 --warmup 2 --regions 1 --no-extras --no-cpu-baseline --seqs 32 --ctxs 1 --host-threads 1 --side-stream off"
cd /tmp
rm -rf $OUT/ic_pmc
timeout 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH GRBM_GUI_ACTIVE --output-format csv -d $OUT/ic_pmc -o p -- python3 $BENCH $ARGS > $OUT/ic_pmc.log 2>&1
cd - > /dev/null
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/ic_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:22]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("GRBM_GUI_ACTIVE", [0, 1])[0])[:8]:
    print(k, {c: round(v[0] / v[1]) for c, v in acc[k].items()})
PY
tail -2 $OUT/ic_pmc.log | cut -c1-200

#!/bin/bash
# What bounds k_klt_track?  Three rocprofv3 --pmc passes (texture-address unit, L1, sequencer) over a one-context bench run.
#   tools/pmc_klt.sh [tag]  -> gpurun_out/<tag>_pmc_klt.txt (mean per k_klt_track launch)
TAG=${1:-k}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
BENCH="$PWD/bench.py"
ARGS="--steps 4 --warmup 2 --regions 1 --no-extras --no-cpu-baseline --seqs 32 --ctxs 1 --host-threads 1 --side-stream off"
cd /tmp
i=0
# at most 2 texture-addresser counters per pass (more: "Request exceeds the capabilities of the hardware" and the run hangs);
# every pass under its own timeout
for CTRS in "TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS GRBM_GUI_ACTIVE" \
            "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" \
            "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TD_TD_BUSY" \
            "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1))
  rm -rf $OUT/${TAG}_pmck$i
  timeout 200 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/${TAG}_pmck$i -o p -- python3 $BENCH $ARGS > $OUT/${TAG}_pmck$i.log 2>&1
done
cd - > /dev/null
python3 - <<PY > $OUT/${TAG}_pmc_klt.txt
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/${TAG}_pmck*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_klt_track" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print("%-34s %14.0f  (mean of %d launches)" % (k, acc[k][0] / acc[k][1], acc[k][1]))
PY
cat $OUT/${TAG}_pmc_klt.txt

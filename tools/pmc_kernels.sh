#!/bin/bash
# Sequencer counters per kernel of one batched context (what each kernel's waves spend their cycles on):
#   tools/pmc_kernels.sh [tag]  -> gpurun_out/<tag>_pmc_kernels.txt
# rocprofv3 --pmc passes only (no trace domains); every pass under its own timeout.
TAG=${1:-k}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
BENCH="$PWD/bench.py"
ARGS="--steps 4 --warmup 2 --regions 1 --no-extras --no-cpu-baseline --seqs 32 --ctxs 1 --host-threads 1 --side-stream off"
cd /tmp
i=0
for CTRS in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rm -rf $OUT/${TAG}_pmcq$i
  timeout 200 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/${TAG}_pmcq$i -o p -- python3 $BENCH $ARGS > $OUT/${TAG}_pmcq$i.log 2>&1
done
cd - > /dev/null
python3 - <<PY > $OUT/${TAG}_pmc_kernels.txt
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/${TAG}_pmcq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:22]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
names = sorted({c for k in acc for c in acc[k]})
for k in sorted(acc, key=lambda k: -acc[k].get("GRBM_GUI_ACTIVE", [0, 1])[0]):
    if not k.startswith("k_"):
        continue
    print(k)
    for c in names:
        if c in acc[k]:
            print("    %-24s %16.0f  (mean of %d launches)" % (c, acc[k][c][0] / acc[k][c][1], acc[k][c][1]))
PY
cat $OUT/${TAG}_pmc_kernels.txt

#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01 ["extra bench args"]
# 1. kernel trace + stats of the default bench command   -> gpurun_out/<tag>_stats
# 2. PMC FETCH_SIZE and WRITE_SIZE in two separate passes -> gpurun_out/<tag>_pmc_{fetch,write}; MFMA counters in a third
# 3. tools/summarize_profiles.py condenses them into gpurun_out/<tag>_* files that are copied to profiles/ by hand.
# The program itself follows `--` (no env / bash -c hops under the profiler).
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
BENCH="$PWD/bench.py"
EXTRA=${2:-}
ARGS="--steps 20 --warmup 5 --regions 1 --no-extras --no-cpu-baseline $EXTRA"
PMCARGS="--steps 4 --warmup 2 --regions 1 --no-extras --no-cpu-baseline $EXTRA"
cd /tmp
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_mfma $OUT/${TAG}_pmc_valu
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o stats -- python3 $BENCH $ARGS > $OUT/${TAG}_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o fetch -- python3 $BENCH $PMCARGS > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o write -- python3 $BENCH $PMCARGS > $OUT/${TAG}_pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc MfmaUtil MfmaFlopsF64 --output-format csv -d $OUT/${TAG}_pmc_mfma -o mfma -- python3 $BENCH $PMCARGS > $OUT/${TAG}_pmc_mfma.log 2>&1
# vector-instruction issue: the roof that binds k_klt_track (bench.py roofline.valu)
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_valu -o valu -- python3 $BENCH $PMCARGS > $OUT/${TAG}_pmc_valu.log 2>&1
cd - > /dev/null
# the same command on ONE stream (per-kernel durations without the other streams' kernels on the chip), then the summaries, then the copies
# under profiles/ that bench.py and profiles/README_<tag>.md refer to
bash tools/kstats256.sh $TAG > /dev/null 2>&1
python3 tools/summarize_profiles.py $TAG
bash tools/collect_profiles.sh $TAG > /dev/null

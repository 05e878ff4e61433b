#!/bin/bash
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
BENCH="$PWD/bench.py"
cd /tmp
rm -rf $OUT/t1_ks
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/t1_ks -o ks -- python3 $BENCH --steps 30 --warmup 5 --regions 1 --no-extras --seqs 1 --ctxs 1 --host-threads 1 --no-cpu-baseline --side-stream ${1:-pipeline} > $OUT/t1_ks.log 2>&1
cd - > /dev/null
tail -1 $OUT/t1_ks.log | cut -c1-200

#!/bin/bash
# A/B old vs new BA kernels on one box: headline value and kstats at batch 32
for v in 0 1 0 1; do
  echo "== VO_BA_V2=$v default bench"
  VO_BA_V2=$v python bench.py --no-extras --no-cpu-baseline --regions 3 --steps 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('ba_lm_iterations_run'), d['config'].get('ba_final_cost'))"
done

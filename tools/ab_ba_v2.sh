#!/bin/bash
# A/B of the two BA kernel families on one box (vo_tuning.ba_kernels: 1 = lane per observation, 2 = wave-private): the headline value
for v in 1 2 1 2; do
  echo "== ba_kernels=$v default bench"
  python bench.py --no-extras --no-cpu-baseline --full-line --regions 3 --steps 50 --tune ba_kernels=$v 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('ba_lm_iterations_run'), d['config'].get('ba_final_cost'))"
done

for cfg in "256 1 1" "256 2 2" "256 4 4" "384 1 1" "512 2 2"; do
  set -- $cfg
  echo "== seqs $1 ctxs $2 threads $3"
  python bench.py --no-extras --no-cpu-baseline --regions 3 --steps 40 --seqs $1 --ctxs $2 --host-threads $3 --side-stream pipeline --full-line --extras-file /dev/null 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done

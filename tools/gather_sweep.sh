# frames from the host: workgroups of k_gather_frames (vo_tuning.gather_workgroups) against the headline's host_frames figure and the closed loop
for g in 0 16 32 64 128 256 512; do
  T=""; [ $g != 0 ] && T="--tune gather_workgroups=$g"
  python3 bench.py --no-cpu-baseline --steps 40 --regions 3 --full-line --extras-file /dev/null $T 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['host_frames']; print('G=$g (0 = the rule) headline', d['value'], 'host', h['value'], h['vs_resident'], h['h2d_gb_s'])"
  python3 bench.py --workload pipeline --pipe-window 10 --pipe-no-resurrect --seqs 256 --ctxs 1 --steps 40 --warmup 10 --regions 3 --no-cpu-baseline --full-line --extras-file /dev/null --pipe-host-frames $T 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('G=$g closed loop host', d['value'], d['ms_per_step'])"
done

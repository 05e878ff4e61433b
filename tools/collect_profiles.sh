#!/bin/bash
# copies the condensed rocprofv3 summaries of a round (tools/profile_round.sh <tag>; tools/kstats256.sh <tag>; tools/summarize_profiles.py <tag>)
# from gpurun_out/ (scratch) into profiles/ (tracked) under the names bench.py and profiles/README_<tag>.md refer to
TAG=${1:-r06}
G=gpurun_out; P=profiles
cp $G/${TAG}_kernel_stats.csv $P/${TAG}_kernel_stats_default.csv
[ -f $G/${TAG}_kernel_stats_one_stream.csv ] && cp $G/${TAG}_kernel_stats_one_stream.csv $P/${TAG}_kernel_stats_one_stream.csv
for c in traffic valu mfma; do cp $G/${TAG}_pmc_$c.csv $P/${TAG}_pmc_${c}_default.csv; done
[ -f $G/${TAG}_kstats256.txt ] && cp $G/${TAG}_kstats256.txt $P/${TAG}_kstats_batch256_single_stream.txt
cp $G/kernel_counters.json $G/klt_traffic.json $G/klt_valu.json $P/
ls -la $P/${TAG}_* $P/kernel_counters.json $P/klt_*.json

#!/bin/bash
# Copy the judged summaries of a gpurun_out/<run> directory (written by bench.py, scripts/prof_bench.sh, scripts/pmc_traffic.sh,
# scripts/layer_table.py, build/tune_pw) into profiles/ under a round tag:  collect_profiles.sh gpurun_out/r2m r02_m
set -eu
SRC=$1; TAG=$2; D=profiles
[ -f $SRC/bench_default.json ] && cp $SRC/bench_default.json $D/${TAG}_bench_default.json
[ -f $SRC/bench_default.log ] && cp $SRC/bench_default.log $D/${TAG}_bench_default.log
for s in one two; do
  if [ -d $SRC/prof_$s ]; then
    cp $SRC/prof_$s/prof/prof_kernel_stats.csv $D/${TAG}_${s}_stream_kernel_stats.csv
    python3 scripts/stats_summary.py $SRC/prof_$s/prof/prof_kernel_stats.csv 5 > $D/${TAG}_${s}_stream_kernel_stats_summary.txt
  fi
done
if [ -d $SRC/pmc ]; then
  python3 scripts/pmc_summary.py --json $D/pmc_traffic.json $SRC/pmc/FETCH_SIZE/pmc_counter_collection.csv $SRC/pmc/WRITE_SIZE/pmc_counter_collection.csv > $D/${TAG}_pmc_hbm_traffic.txt
fi
for f in layer_table.log lt_policy.log variants.log tune_pw128.log; do [ -f $SRC/$f ] && cp $SRC/$f $D/${TAG}_$f; done
ls -la $D | grep $TAG

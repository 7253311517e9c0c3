#!/bin/bash
# rocprofv3 kernel statistics of a short bench run: prof_bench.sh <outdir> [bench args...]   (5 steps profiled: 1 warm-up + 4)
# (--no-bn-calibration: the one-off BatchNorm calibration pass of bench.py would add 106 small forward launches to the per-kernel
#  averages; kernel times do not depend on the weights)
set -u
OUT=$1; shift; R=$GRAFT_REPO_ROOT; mkdir -p $R/$OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o prof --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline --no-bn-calibration --lr 1e-7 "$@" > $R/$OUT/bench.log 2>&1
python3 $R/scripts/stats_summary.py $(ls $R/$OUT/prof/*kernel_stats.csv | head -1) 5 > $R/$OUT/kernel_stats_summary.txt 2>&1

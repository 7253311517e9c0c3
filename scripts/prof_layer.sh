#!/bin/bash
# rocprofv3 kernel statistics of scripts/layer_table.py for the rows in CXRK_LAYER_ROWS: prof_layer.sh <outdir> [batch] [reps]
set -u
OUT=$1; shift; R=$GRAFT_REPO_ROOT; mkdir -p $R/$OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o prof --output-format csv -- python3 $R/scripts/layer_table.py "$@" > $R/$OUT/layer.log 2>&1
python3 $R/scripts/stats_summary.py $(ls $R/$OUT/prof/*kernel_stats.csv | head -1) 1 > $R/$OUT/kernel_stats_summary.txt 2>&1

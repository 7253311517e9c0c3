#!/bin/bash
# HBM traffic per kernel from the PMC counters (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE in SEPARATE
# passes with --kernel-trace only, over one bench step.  pmc_traffic.sh <outdir>; then scripts/pmc_summary.py builds the table
# and profiles/pmc_traffic.json (gfx950 correction: HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024).
set -u
OUT=$1; R=$GRAFT_REPO_ROOT; mkdir -p $R/$OUT
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $R/$OUT/$c -o pmc --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline --no-bn-calibration --lr 1e-7 > $R/$OUT/$c.log 2>&1 || echo "pass $c failed"
done

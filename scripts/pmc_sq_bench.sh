#!/bin/bash
# MFMA-busy and occupancy counters of the step's OWN kernels: SQ / GRBM counter passes over one bench step (the program directly
# after `--`, counters in their own run with --kernel-trace only).  pmc_sq_bench.sh <outdir>; then
#   python3 scripts/pmc_sq_summary.py --json profiles/pmc_sq.json <outdir>/p1/pmc_counter_collection.csv [<outdir>/p2/...] > profiles/rNN_pmc_sq_step.txt
set -u
OUT=$1; R=$GRAFT_REPO_ROOT; mkdir -p $R/$OUT
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline --no-bn-calibration --lr 1e-7"
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $R/$OUT/p1 -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $R/$OUT/p1.log 2>&1 || echo "pass 1 failed"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace -d $R/$OUT/p2 -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $R/$OUT/p2.log 2>&1 || echo "pass 2 failed"
find $R/$OUT -name "*counter_collection.csv" | head

#!/bin/bash
# SQ counter passes (8 slots each) over a standalone binary: pmc_sq.sh <binary> <outdir>
set -u
BIN=$1; OUT=$2; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $R/$OUT/p1 -o p1 --output-format csv -- $R/$BIN > $R/$OUT/p1.log 2>&1 || echo "pass1 failed"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --kernel-trace -d $R/$OUT/p2 -o p2 --output-format csv -- $R/$BIN > $R/$OUT/p2.log 2>&1 || echo "pass2 failed"
rocprofv3 -L > $R/$OUT/counters.txt 2>&1 || true

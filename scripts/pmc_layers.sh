#!/bin/bash
# SQ instruction-mix counters per kernel over the layer table (batch 256, 1 rep): pmc_layers.sh <outdir>
set -u
OUT=$1; R=$GRAFT_REPO_ROOT; mkdir -p $R/$OUT
cd /tmp; export TMPDIR=/tmp
export CXRK_PRECISION=split_bf16
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $R/$OUT/p1 -o p1 --output-format csv -- python3 $R/scripts/layer_table.py 256 1 > $R/$OUT/p1.log 2>&1 || echo "pass1 failed"

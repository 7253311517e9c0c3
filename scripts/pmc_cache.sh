#!/bin/bash
# L2 hit / miss and fabric request counters over a standalone binary: pmc_cache.sh <binary> <outdir>
set -u
BIN=$1; OUT=$2; R=$GRAFT_REPO_ROOT; mkdir -p $R/$OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d $R/$OUT/c1 -o c1 --output-format csv -- $R/$BIN > $R/$OUT/c1.log 2>&1 || echo "pass c1 failed"
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace -d $R/$OUT/c2 -o c2 --output-format csv -- $R/$BIN > $R/$OUT/c2.log 2>&1 || echo "pass c2 failed"

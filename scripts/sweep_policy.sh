#!/bin/bash
# Step time of bench.py under tile / split-K policy overrides, all on ONE box (boxes differ by +-1.5 %): sweep_policy.sh <outfile>
OUT=$1; R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $(dirname $R/$OUT); : > $R/$OUT
run() {  # run <label> [ENV=VALUE...]
  label=$1; shift
  ms=$(env "$@" python $R/bench.py --no-secondary --no-cpu-baseline --no-roofline --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f %s' % (d['ms_per_step'], ' '.join('%.1f' % x for x in d['step_ms'])))")
  echo "$label: $ms" | tee -a $R/$OUT
}
run default CXRK_DUMMY=0
run wgrad_blocks_512 CXRK_WGRAD_BLOCKS=512
run wgrad_blocks_384 CXRK_WGRAD_BLOCKS=384
run wgrad_blocks128_1024 CXRK_WGRAD_BLOCKS128=1024
run wgrad_blocks128_2048 CXRK_WGRAD_BLOCKS128=2048
run mink_fprop_256 CXRK_MINK_FPROP=256
run mink_plain_256 CXRK_MINK_PLAIN=256
run wide_eff_75 CXRK_WIDE_EFF=75
run wide_eff_45 CXRK_WIDE_EFF=45
run nt_16mb CXRK_NT_MB=16
run nt_1024mb CXRK_NT_MB=1024
run one_stream CXRK_TWO_STREAMS=0
run default_again CXRK_DUMMY=0

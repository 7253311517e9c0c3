#!/bin/bash
# BASELINE config 4 (CLASS_INCREMENTAL class-pos-neg, 5-label sequence, global batch 2048 on 4 GPUs) rehearsed with 4 ranks on ONE
# GPU: gloo transport (RCCL refuses several ranks per device; same collectives, same code path), every rank on device 0.
#   rehearse_cfg4.sh <outdir> [adapter|joint]
# adapter: the reference's step (adapters on pre-computed embeddings), gradients all-reduced;  joint: both encoders in-loop, the
# shards' embeddings all-gathered for the global 2048 x 2048 similarity matrix, gradient ranges all-reduced from inside the backward.
set -u
OUT=$1; KIND=${2:-adapter}; R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/$OUT
export CXRK_DIST_BACKEND=gloo CXRK_DRIVER_DEVICE=0 CXRK_PRECISION=split_bf16 HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=4
ARGS="class-inc --mode class-pos-neg --more-labels --batch-size 2048 --epochs 1 --log-root /tmp/cfg4_runs_$KIND"   # (checkpoints of the joint form are 0.5 GB: keep them out of gpurun_out)
if [ "$KIND" = joint ]; then ARGS="$ARGS --joint --n-train 10240 --n-eval 256 --lr 1e-6"; else ARGS="$ARGS --n-train 61440 --n-eval 4096"; fi
cd $R
python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29511 \
    -m incremental_multimodal_medical_learning_ii_amd.drivers $ARGS
python - <<PY
import glob, json
for f in sorted(glob.glob("/tmp/cfg4_runs_$KIND/*/scalars.jsonl")):
    rows = [json.loads(l) for l in open(f)]
    loss = [round(r["value"], 5) for r in rows if r["tag"] == "train/Loss"]
    acc = [round(r["value"], 4) for r in rows if r["tag"] == "test/Accuracy"]
    print(f.split("/")[-2], "train/Loss", loss, "test/Accuracy", acc)
PY

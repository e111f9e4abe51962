#!/bin/bash
# Run ON the GPU box: training timelines (bf16, fp16x3) of the current library; prints the conv2-dgrad / wgrad0 / reduce lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/gpu_timeline.sh bf16 fp16x3
for p in bf16 fp16x3; do echo "== $p"; grep -n "Lb0ELi2ELb1EEv8\|false, 2, true\|wgrad0\|reduce_rows\|step span" gpurun_out/timeline_$p.txt | cut -c1-140; done

#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for p in bf16 fp16x3; do
timeout 600 python tools/probes/train_ab.py $p bwd_prefill=0 bwd_prefill=1 bwd_prefill=16 bwd_prefill=48 deterministic_backward=1,bwd_prefill=1 deterministic_backward=0,bwd_prefill=1 steps=8 rounds=4 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r6_ab_prefill.txt
echo "== formal fence A/B (LM forward + backward kernels)"
for prec in bf16; do VARIANTS_PRECISION=$prec VARIANTS_TRAIN=6 python tools/ab_libs.py libhla.so libhla_fence.so libhla.so libhla_fence.so 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r6_ab_fence.txt

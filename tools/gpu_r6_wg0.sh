#!/bin/bash
# Run ON the GPU box: round-6 fused conv0 weight gradient -- its test, the gradient tests, same-process A/B against the stored-map
# path and the 2-workgroup build of conv2's data gradient.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_conv0 or train_step_gradients or wave_specialised or backward_dynamic_trimming or ford_train_step or two_rank_real" 2>&1 | grep -v "^$" | tail -15
for p in bf16 fp16x3; do
timeout 600 python tools/probes/train_ab.py $p wgrad_two_phase=0 wgrad_two_phase=2 steps=8 rounds=4 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r6_ab_wg0.txt
for prec in bf16 fp16x3; do VARIANTS_PRECISION=$prec VARIANTS_TRAIN=6 python tools/ab_libs.py libhla.so libhla_occ2.so libhla.so libhla_occ2.so 2>&1 | grep -v amdgpu.ids | grep train; done | tee gpurun_out/r6_ab_occ2.txt

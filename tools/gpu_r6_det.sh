#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deterministic_backward or lm_backward_small" 2>&1 | grep -v "^$" | tail -3
for p in bf16 fp16x3; do
timeout 600 python tools/probes/train_ab.py $p deterministic_backward=0 deterministic_backward=1 steps=8 rounds=4 2>&1 | grep -v amdgpu.ids | tail -3
done | tee gpurun_out/r6_ab_det.txt

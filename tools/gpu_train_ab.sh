#!/bin/bash
# Run ON the GPU box: gpurun -- 'bash tools/gpu_train_ab.sh "<pytest -k expr>" libA.so libB.so ...'   same-box A/B of the TRAINING step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
K=$1; shift
if [ -n "$K" ]; then timeout 1500 python -m pytest tests -q -m gpu -x -k "$K" > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/gpu_tests.log | tail -3; grep -B5 -A25 "^E " gpurun_out/gpu_tests.log | head -60; fi
for rep in 1 2; do for prec in fp16x3 bf16; do for lib in "$@"; do echo -n "$prec "; VARIANTS_TRAIN=4 VARIANTS_PRECISION=$prec python tools/ab_libs.py $lib 2>&1 | head -1; done; done; done | tee gpurun_out/ab_train.log

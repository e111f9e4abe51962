#!/bin/bash
# Run ON the GPU box: gpurun -- 'bash tools/gpu_ab.sh "<pytest -k expr or empty for all>" libA.so libB.so ...'
# parity tests first (a fast kernel with different results is not done), then a same-box A/B of the listed library builds.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
K=$1; shift
timeout 1500 python -m pytest tests -q -m gpu -x ${K:+-k "$K"} > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/gpu_tests.log | tail -3
grep -B5 -A25 "^E " gpurun_out/gpu_tests.log | head -80
for prec in bf16 fp16x3; do echo "== $prec"; VARIANTS_PRECISION=$prec python tools/ab_libs.py "$@" 2>&1; done | tee gpurun_out/ab.log

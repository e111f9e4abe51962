#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vgg or conv0 or e2e_kitti_full or determinism or dead_ground or e2e_ford_full or train_step_gradients or hires" 2>&1 | grep -v "^$" | tail -3
for prec in bf16 fp16x3; do VARIANTS_PRECISION=$prec python tools/ab_libs.py libhla_occ2.so libhla.so libhla_c02p.so libhla_occ2.so libhla.so libhla_c02p.so 2>&1 | grep -v amdgpu.ids | cut -c1-330; done | tee gpurun_out/r6_ab_c02.txt

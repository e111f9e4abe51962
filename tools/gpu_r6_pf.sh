#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_conv0 or train_step or wave_specialised or backward_dynamic_trimming or ford_train or two_rank_real or level4 or g2s_train or vgg_backward or gradient_fidelity" 2>&1 | grep -v "^$" | tail -4
for prec in bf16 fp16x3; do VARIANTS_PRECISION=$prec VARIANTS_TRAIN=6 python tools/ab_libs.py libhla_nopf.so libhla.so libhla_nopf.so libhla.so 2>&1 | grep -v amdgpu.ids | grep train | cut -c1-300; done | tee gpurun_out/r6_ab_pf.txt
HLA_LIB=$PWD/highlyaccurate_amd/libhla_stamps.so HLA_ALLOW_STALE=1 timeout 600 python tools/probes/conv_stamps_train.py bf16 29 41 28 40 19 2>&1 | grep -v amdgpu | cut -c1-330

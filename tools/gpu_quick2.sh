cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vgg_backward or train_step or backward_row or ground_crop or standalone or two_rank or determinism or e2e_kitti_full" 2>&1 | tail -4
VARIANTS_TRAIN=4 python tools/variants.py libhla_base.so libhla.so libhla_base.so libhla.so 2>&1 | tail -8

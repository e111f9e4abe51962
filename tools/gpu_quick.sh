cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vgg_small or vgg_level4 or split_fp16 or e2e_kitti_full_shape or e2e_kitti_features or vgg_backward_small or train_step_gradients or hires or level4_train or standalone" 2>&1 | grep -v "^$" | tail -4
for p in "$@"; do timeout 300 python bench.py --precision $p --train-steps 0 --no-cpu-baseline --steps 20 --warmup 5 --no-extra-legs > gpurun_out/b_$p.json 2> gpurun_out/b_$p.err; python - <<PY
import json
r=json.load(open('gpurun_out/b_$p.json')); print('$p', r['value'], r['ms_per_step'], {k:(v['avg_us'],v['tflops']) for k,v in r['kernels'].items() if 'conv' in k})
PY
done

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vgg_small or vgg_level4 or e2e_kitti_full_shape or dead_ground or determinism or vgg_backward_small or train_step_gradients or ragged" 2>&1 | tail -5
python tools/variants.py libhla.so 2>&1 | tail -3
for p in fp16x3; do timeout 300 python bench.py --precision $p --train-steps 0 --no-cpu-baseline --steps 20 --warmup 5 --no-extra-legs > gpurun_out/b_$p.json 2> gpurun_out/b_$p.err; python - <<PY
import json
r=json.load(open('gpurun_out/b_$p.json')); print('$p', r['value'], r['ms_per_step'], {k:(v['avg_us'],v['tflops']) for k,v in r['kernels'].items() if 'conv' in k})
PY
done

# Run ON the GPU box (gpurun -- bash tools/gpu_trim.sh): the backward-trimming tests, then the training leg dense vs trimmed (A/B).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "dynamic_trimming or row_trimming or train_step_gradients or vgg_backward or two_rank or g2s_train_step or ford_train_step" 2>&1 | grep -v "^$" | grep "dynamic trimming\|row trimming\|passed\|failed\|Error\|error" | tail -30
for dense in 1 0; do
HLA_VGG_BWD_DENSE=$dense timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --no-extra-legs --no-kernel-timing --train-steps 8 > gpurun_out/bt_$dense.json 2> gpurun_out/bt_$dense.err; python - <<PY
import json
r=json.load(open('gpurun_out/bt_$dense.json')); print('dense=$dense', r['value'], {k: v for k, v in r.items() if 'train' in k})
PY
done

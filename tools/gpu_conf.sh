# Run ON the GPU box (gpurun -- bash tools/gpu_conf.sh): confidence-head tests, their per-launch time, one training-leg bench.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vgg_small or vgg_level4 or confidence or using_weight or ragged or train_mode_forward" 2>&1 | tail -3
python tools/probes/train_launches.py 2>&1 | grep "conf_kernel\|total kernel"
HLA_VGG_BWD_DENSE=0 timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --no-extra-legs --no-kernel-timing --train-steps 8 > gpurun_out/bt.json 2> gpurun_out/bt.err; python - <<PY
import json
r=json.load(open('gpurun_out/bt.json')); print(r['value'], {k: v for k, v in r.items() if 'train' in k})
PY

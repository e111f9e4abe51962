#!/bin/bash
# Run ON the GPU box: gradient parity tests, then the training step (per-launch exclusive table + the timed leg) in the given precisions
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
K=${K:-"train_step_gradients or gradient_fidelity or backward_row_trimming or backward_dynamic_trimming or vgg_backward_small or training_ground_crop"}
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" 2>&1 | tail -4
for p in "$@"; do
  python tools/probes/train_launches.py $p 0 > gpurun_out/train_excl_$p.txt 2>&1; grep -A8 "total kernel ms" gpurun_out/train_excl_$p.txt
  python bench.py --precision $p --train-precision $p --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-kernel-timing --train-steps 6 > gpurun_out/tq_$p.json 2>gpurun_out/tq_$p.err
  python - <<PY
import json
r=json.load(open('gpurun_out/tq_$p.json')); print('train $p', r['train']['value'], r['train']['blocks_ms_per_step'], r['train'].get('with_train_ground_crop'))
PY
done

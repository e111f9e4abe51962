#!/bin/bash
# Run ON the GPU box: round-6 LM-backward changes -- targeted tests, the op log, a train-only bench in both modes and timelines.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lm_backward_small or deterministic_backward or zero_fill or pose_loss or train_step_gradients or ford_train_step or ablation_optimisers_backward or two_rank_real or g2s_lm_backward or training_steps_do_not or wave_specialised or using_weight" 2>&1 | grep -v "^$" | tail -15
timeout 200 python tools/probes/step_dispatch.py bf16 4 > gpurun_out/r6_dispatch.txt 2>&1; tail -40 gpurun_out/r6_dispatch.txt
for p in bf16 fp16x3; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --precision $p --train-precision $p --train-steps 6 > gpurun_out/r6_train_$p.json 2> gpurun_out/r6_train_$p.err
  python - <<PY
import json
r = json.load(open('gpurun_out/r6_train_$p.json')); t = r['train']
print('$p', 'train', t.get('value'), t.get('ms_per_step'), t.get('blocks_ms_per_step'), (t.get('roofline') or {}).get('frac'), t.get('error'))
print('  breakdown', {k: v for k, v in (t.get('step_breakdown') or {}).items() if k != 'what'})
PY
done
bash tools/gpu_timeline.sh bf16
grep -n "pose_loss\|lm_bwd_solve\|zero_fill\|FillFunctor\|rocclr" gpurun_out/timeline_bf16.txt | head -40
tail -3 gpurun_out/timeline_bf16.txt

#!/bin/bash
# Run ON the GPU box: round-6 first look -- the default bench line, which host op launches what in a training step, and the
# training timeline (with the inter-step gap) in both training modes.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.err; echo bench rc $?
python - <<'PY'
import json
r = json.load(open('gpurun_out/r6_bench.json'))
print('value', r['value'], r['ms_per_step'], 'frac', r['roofline']['frac'])
print('matched', r.get('matched_accuracy'))
t = r['train']; print('train', t.get('value'), t.get('ms_per_step'), t.get('blocks_ms_per_step'), t.get('roofline', {}).get('frac'))
print('breakdown', t.get('step_breakdown'))
for p, e in (t.get('by_precision') or {}).items(): print(' ', p, e.get('value'), e.get('ms_per_step'), (e.get('roofline') or {}).get('frac'))
for k, e in (r.get('secondary') or {}).items(): print(' ', k, e.get('value'), e.get('ms_per_step'), (e.get('accuracy') or {}).get('final_pose_dev_shift_m'), e.get('error'))
print('tele', r['roofline'].get('telemetry'))
PY
timeout 300 python tools/probes/step_ops.py bf16 32 > gpurun_out/r6_step_ops_bf16.txt 2>&1; tail -45 gpurun_out/r6_step_ops_bf16.txt
bash tools/gpu_timeline.sh bf16 fp16x3
tail -4 gpurun_out/timeline_bf16.txt gpurun_out/timeline_fp16x3.txt

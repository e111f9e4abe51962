#!/bin/bash
# Run ON the GPU box:  gpurun --timeout 1500 -- 'bash tools/gpu_round.sh [pytest -k expression]'
# Full `pytest -m gpu` (printed deviations kept: -rP), the default bench line, the single-pair latency.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
K=${1:-}
timeout 1500 python -m pytest tests -q -m gpu -x -rP --durations=12 ${K:+-k "$K"} > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/gpu_tests.log | tail -3
grep -E "^(B=32|dead-row|backward|2-rank|hires|ford|fuzz|smoke:|reduced|level|train)" gpurun_out/gpu_tests.log | head -60
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    r = json.load(open('gpurun_out/bench.json'))
    print('value', r['value'], 'ms', r['ms_per_step'], 'roofline', r['roofline']['achieved'], r['roofline']['frac'])
    print({k: (v['avg_us'], v['tflops']) for k, v in r['kernels'].items()})
    print('by_precision', {k: v.get('value') for k, v in r['by_precision'].items()})
    print('secondary', {k: v.get('value') for k, v in r['secondary'].items()})
    print('train', {k: v for k, v in r['train'].items() if k != 'kernels'})
    print('cpu', r['cpu_baseline'])
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/bench.err').read()[-1500:])
PY
timeout 300 python bench.py --batch 1 --steps 300 --warmup 30 --no-extra-legs --train-steps 0 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('B=1 latency ms', r['ms_per_step'])"

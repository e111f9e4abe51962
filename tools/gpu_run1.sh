set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "vgg_small or vgg_level4 or split_fp16 or e2e_kitti_full_shape or e2e_kitti_features or reduced_precision" 2>&1 | grep -v "^$" | tail -60 > gpurun_out/t1.log
tail -40 gpurun_out/t1.log
for p in fp16x3 fp32 bf16; do timeout 300 python bench.py --precision $p --train-steps 0 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/b_$p.json 2> gpurun_out/b_$p.err; python - <<PY
import json
try:
    r=json.load(open('gpurun_out/b_$p.json')); print('$p', r['value'], r['ms_per_step'], r.get('roofline'), {k:(v['avg_us'],v['tflops']) for k,v in r['kernels'].items()})
except Exception as e: print('$p failed', e); print(open('gpurun_out/b_$p.err').read()[-2000:])
PY
done

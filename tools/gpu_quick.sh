cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "vgg_small or lm_solve or e2e_kitti_full_shape or e2e_ford_full or dead_ground or determinism or reduced_precision or full_bench or ragged or error_behaviour or variants_vs_golden or hires" 2>&1 | grep -v "^$" | tail -25
for p in bf16 fp16x3; do timeout 300 python bench.py --precision $p --train-steps 0 --no-cpu-baseline --steps 30 --warmup 5 --no-extra-legs > gpurun_out/b_$p.json 2> gpurun_out/b_$p.err; python - <<PY
import json
r=json.load(open('gpurun_out/b_$p.json')); print('$p', r['value'], r['ms_per_step'], {k:(v['avg_us'],v['tflops']) for k,v in r['kernels'].items()})
PY
done

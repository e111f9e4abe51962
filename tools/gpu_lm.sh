cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lm_solve or e2e_kitti_full_shape or e2e_ford_full or determinism or reduced_precision or ragged or error_behaviour or variants_vs_golden or train_mode_forward or lm_backward_small" 2>&1 | grep -v "^$" | tail -4
for p in bf16 fp16x3; do for f in 1 0; do HLA_LM_FEAT16=$f timeout 300 python bench.py --precision $p --train-steps 0 --no-cpu-baseline --steps 30 --warmup 5 --no-extra-legs > gpurun_out/b_$p.json 2> gpurun_out/b_$p.err; python - <<PY
import json
r=json.load(open('gpurun_out/b_$p.json')); print('$p feat16=$f', r['value'], r['ms_per_step'], {k:v['avg_us'] for k,v in r['kernels'].items() if 'lm' in k})
PY
done; done

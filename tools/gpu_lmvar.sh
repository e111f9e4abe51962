cd $GRAFT_REPO_ROOT
for lib in libhla.so libhla_v140.so libhla_v141.so libhla_v142.so libhla.so; do for p in bf16 fp16x3; do HLA_LIB=$PWD/highlyaccurate_amd/$lib timeout 300 python bench.py --precision $p --train-steps 0 --no-cpu-baseline --steps 20 --warmup 5 --no-extra-legs > gpurun_out/b.json 2> gpurun_out/b.err; python - <<PY
import json
r=json.load(open('gpurun_out/b.json')); print('$lib $p', r['value'], r['ms_per_step'], {k:v['avg_us'] for k,v in r['kernels'].items() if 'lm' in k})
PY
done; done

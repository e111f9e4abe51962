# Run ON the GPU box: training-leg time and lm_bwd_accum launch time of experiment builds, e.g. bash tools/gpu_lmbwdvar.sh libhla.so libhla_v156.so
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
HLA_ALLOW_STALE=1 HLA_LIB=$GRAFT_REPO_ROOT/highlyaccurate_amd/$lib timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-extra-legs --train-steps 8 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); t=r['train']
print('$lib', t['value'], t['ms_per_step'], {n:v['avg_us'] for n,v in t['kernels'].items() if n.startswith('lm_')})"
done

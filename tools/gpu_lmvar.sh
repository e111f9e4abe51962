# Run ON the GPU box: LM accumulate kernel times of experiment builds (HLA_LIB), e.g. bash tools/gpu_lmvar.sh libhla.so libhla_v150.so
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
HLA_ALLOW_STALE=1 HLA_LIB=$GRAFT_REPO_ROOT/highlyaccurate_amd/$lib timeout 300 python bench.py --train-steps 0 --no-cpu-baseline --steps 20 --warmup 5 --no-extra-legs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r['kernels']
print('$lib', r['value'], {n:v['avg_us'] for n,v in k.items() if n.startswith('lm_')})"
done

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; echo rc=$?
tail -5 gpurun_out/bench_full.err
python - <<PY
import json
r=json.load(open('gpurun_out/bench_full.json'))
for k in ('value','ms_per_step','roofline','lm_roofline','by_precision','secondary','cpu_baseline'):
    print(k, json.dumps(r.get(k), indent=None)[:1500])
print('train', {k:v for k,v in r['train'].items() if k!='kernels'})
PY

#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for l in libhla_stamps.so libhla_stamps_e1.so; do
HLA_LIB=$PWD/highlyaccurate_amd/$l HLA_ALLOW_STALE=1 timeout 600 python tools/probes/conv_stamps.py bf16 > gpurun_out/stamps_$l.json 2> gpurun_out/stamps_$l.err
python - <<PY
import json
d = json.load(open('gpurun_out/stamps_$l.json'))
for r in d['launches'][:9]:
    c = r['cycles_mean']
    print('$l', r['layer'], c, r['mfma_duty_per_simd'])
PY
done

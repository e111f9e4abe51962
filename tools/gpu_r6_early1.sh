#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/ab_libs.py libhla.so libhla_early1.so libhla.so libhla_early1.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_ab_early1.txt
for l in libhla.so libhla_early1.so; do
HLA_LIB=$PWD/highlyaccurate_amd/$l HLA_ALLOW_STALE=1 python tools/probes/occupancy_table.py bf16 2>/dev/null > gpurun_out/per_layer_$l.json
python - <<PY
import json
d = json.load(open('gpurun_out/per_layer_$l.json'))
print('$l', ' '.join(f"{r['branch'][0]}.{r['layer']}:{r['us']:.0f}" for r in d['launches']))
PY
done

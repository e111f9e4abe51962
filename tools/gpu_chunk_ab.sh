#!/bin/bash
# Run ON the GPU box: same-box A/B of the chunked high-resolution chain (HLA_VGG_CHUNK = samples per launch; 0 = whole batch)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vgg_small or e2e_kitti_full_shape or determinism or bench_batch_per_sample" 2>&1 | tail -3
for rep in 1 2; do for p in bf16 fp16x3; do for c in 0 4 8 16; do
  HLA_VGG_CHUNK=$c python bench.py --precision $p --train-steps 0 --no-cpu-baseline --steps 30 --warmup 8 --no-extra-legs --no-kernel-timing > gpurun_out/ck.json 2>gpurun_out/ck.err
  python - <<PY
import json
r=json.load(open('gpurun_out/ck.json')); print('$p chunk $c rep $rep', r['value'], r['ms_per_step'])
PY
done; done; done
for c in 0 8; do echo "== per layer chunk $c"; HLA_VGG_CHUNK=$c python tools/probes/infer_launches.py 2>&1 | grep -v amdgpu.ids | tail -12; done
for p in fp16x3 bf16; do for c in 0 8; do
  HLA_VGG_CHUNK=$c python bench.py --precision $p --train-precision $p --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-kernel-timing --train-steps 6 > gpurun_out/ck.json 2>gpurun_out/ck.err
  python - <<PY
import json
r=json.load(open('gpurun_out/ck.json')); print('train $p chunk $c', r['train']['value'], r['train']['blocks_ms_per_step'])
PY
done; done

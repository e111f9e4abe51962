#!/bin/bash
# Run ON the GPU box: per-launch tables of the training step (both modes) and the inference step, plus a kernel-trace timeline.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for p in fp16x3 bf16; do python tools/probes/train_launches.py $p > gpurun_out/train_launches_$p.txt 2>&1; tail -25 gpurun_out/train_launches_$p.txt; done
python tools/probes/infer_launches.py > gpurun_out/infer_launches.txt 2>&1; tail -30 gpurun_out/infer_launches.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
for p in fp16x3 bf16; do
rm -rf /tmp/tl_$p; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-extra-legs --precision $p --train-precision $p --train-steps 2 > $R/gpurun_out/tl_$p.json 2> $R/gpurun_out/tl_$p.err
python $R/tools/probes/train_timeline.py /tmp/tl_$p > $R/gpurun_out/timeline_$p.txt 2>&1; tail -3 $R/gpurun_out/timeline_$p.txt
done

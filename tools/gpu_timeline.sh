#!/bin/bash
# Run ON the GPU box: kernel-trace timeline of one training step in the given precisions -> gpurun_out/timeline_<p>.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
R=$PWD; cd /tmp && export TMPDIR=/tmp
for p in "$@"; do
rm -rf /tmp/tl_$p; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-extra-legs --precision $p --train-precision $p --train-steps 2 > $R/gpurun_out/tl_$p.json 2> $R/gpurun_out/tl_$p.err
python $R/tools/probes/train_timeline.py /tmp/tl_$p > $R/gpurun_out/timeline_$p.txt 2>&1; tail -1 $R/gpurun_out/timeline_$p.txt
done

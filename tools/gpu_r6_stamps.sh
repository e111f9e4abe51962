#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for p in ${@:-bf16 fp16x3}; do
HLA_LIB=$PWD/highlyaccurate_amd/libhla_stamps.so HLA_ALLOW_STALE=1 timeout 600 python tools/probes/conv_stamps.py $p > gpurun_out/r06_conv_cycle_table_$p.json 2> gpurun_out/stamps_$p.err
grep -v amdgpu.ids gpurun_out/stamps_$p.err | grep conv02 | cut -c1-700
done

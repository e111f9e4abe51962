cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "$@"; do
echo "== variant $v"
HLA_BENCH_NOCHECK=1 HLA_LIB=$PWD/highlyaccurate_amd/libhla_v$v.so timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --train-steps 0 --no-extra-legs --no-kernel-timing > gpurun_out/v$v.json 2> gpurun_out/v$v.err
grep "\[dbg\]" gpurun_out/v$v.err | head -9
done

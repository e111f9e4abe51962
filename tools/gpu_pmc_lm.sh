cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pmc_lm; R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_lm/a -- python $R/bench.py --steps 2 --warmup 1 --no-kernel-timing --no-cpu-baseline --train-steps 0 --no-extra-legs > /dev/null 2> $R/gpurun_out/pmc_lm/err_a.txt
rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_lm/b -- python $R/bench.py --steps 2 --warmup 1 --no-kernel-timing --no-cpu-baseline --train-steps 0 --no-extra-legs > /dev/null 2> $R/gpurun_out/pmc_lm/err_b.txt
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_lm/c -- python $R/bench.py --steps 2 --warmup 1 --no-kernel-timing --no-cpu-baseline --train-steps 0 --no-extra-legs > /dev/null 2> $R/gpurun_out/pmc_lm/err_c.txt
cd $R
for d in a b c; do python tools/pmc_summary.py gpurun_out/pmc_lm/$d | grep -A9 "lm_accum<64\|lm_accumILi64" | head -11; done
tail -3 gpurun_out/pmc_lm/err_a.txt
rm -rf gpurun_out/pmc_lm/a gpurun_out/pmc_lm/b gpurun_out/pmc_lm/c

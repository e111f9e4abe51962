#!/bin/bash
# Run ON the GPU box (via gpurun) from the repo root: regenerates everything that gets committed under profiles/.
#   gpurun -- 'bash tools/make_profiles.sh r01'     then copy gpurun_out/profiles_<tag>/* into profiles/
set -x
TAG=${1:-r01}
ROOT=$PWD
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --train-steps 0 > /dev/null 2>> $OUT/bench.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --train-steps 0 > /dev/null 2>> $OUT/bench.err
python $ROOT/tools/pmc_hbm.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_hbm_traffic.json > /dev/null
cp $OUT/${TAG}_pmc_hbm_traffic.json $ROOT/profiles/${TAG}_pmc_hbm_traffic.json   # bench.py reads roofline.traffic from here
python $ROOT/bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $OUT/${TAG}_bench_under_rocprof.json 2>> $OUT/bench.err
cp $OUT/stats/*/*kernel_stats.csv $OUT/${TAG}_rocprofv3_kernel_stats.csv
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --train-steps 0 > /dev/null 2>> $OUT/bench.err
python $ROOT/tools/pmc_summary.py $OUT/pmc_sq > $OUT/${TAG}_pmc_sq_counters.txt
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
ls -la $OUT

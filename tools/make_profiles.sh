#!/bin/bash
# Run ON the GPU box (via gpurun) from the repo root: regenerates everything that gets committed under profiles/.
#   gpurun -- 'bash tools/make_profiles.sh r03'     then copy gpurun_out/profiles_<tag>/* into profiles/
# Every bench invocation below is the headline leg only (--no-extra-legs --train-steps 0: BASELINE configs[1], B = 32, bf16), so
# that a kernel's average duration in a CSV is directly comparable with roofline.avg_launch_us of the JSON printed by that very run.
set -x
TAG=${1:-r05}
ROOT=$PWD
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HEAD="--no-cpu-baseline --train-steps 0 --no-extra-legs"
# HBM traffic: separate --pmc passes (MI355X_MICROARCH.md, HBM section); stamped with the kernel sources' content hash
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $ROOT/bench.py --steps 2 --warmup 1 --no-kernel-timing $HEAD > /dev/null 2>> $OUT/bench.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $ROOT/bench.py --steps 2 --warmup 1 --no-kernel-timing $HEAD > /dev/null 2>> $OUT/bench.err
python $ROOT/tools/pmc_hbm.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_hbm_traffic.json > /dev/null
cp $OUT/${TAG}_pmc_hbm_traffic.json $ROOT/profiles/${TAG}_pmc_hbm_traffic.json   # bench.py reads roofline.traffic from here
# the same two passes around the TRAINING step in the parity-grade mode (wgrad / dgrad / lm_bwd_accum kernels): VERDICT r04 #1a
TR="--steps 1 --warmup 0 --no-kernel-timing --no-cpu-baseline --no-extra-legs --precision fp16x3 --train-precision fp16x3 --train-steps 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_tr -- python $ROOT/bench.py $TR > /dev/null 2>> $OUT/bench.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_tr -- python $ROOT/bench.py $TR > /dev/null 2>> $OUT/bench.err
python $ROOT/tools/pmc_hbm.py $OUT/pmc_fetch_tr $OUT/pmc_write_tr $OUT/${TAG}_pmc_hbm_traffic_train.json > /dev/null
cp $OUT/${TAG}_pmc_hbm_traffic_train.json $ROOT/profiles/${TAG}_pmc_hbm_traffic_train.json
# per launch of both branches: workgroups, resident generations, tail bound, microseconds
python $ROOT/tools/probes/occupancy_table.py bf16 > $OUT/${TAG}_per_layer.json 2>> $OUT/bench.err
# the full default line (by_precision, secondary, train, cpu_baseline)
python $ROOT/bench.py > $OUT/${TAG}_bench.json 2>> $OUT/bench.err
# the judged summary: per-kernel time of the headline leg
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --steps 50 --warmup 10 $HEAD > $OUT/${TAG}_bench_under_rocprof.json 2>> $OUT/bench.err
cp $OUT/stats/*/*kernel_stats.csv $OUT/${TAG}_rocprofv3_kernel_stats.csv
rm -rf $OUT/stats
# the matched-accuracy mode (precision fp16x3) on its own
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --precision fp16x3 --steps 20 --warmup 5 $HEAD > $OUT/${TAG}_bench_fp16x3_under_rocprof.json 2>> $OUT/bench.err
cp $OUT/stats/*/*kernel_stats.csv $OUT/${TAG}_rocprofv3_kernel_stats_fp16x3.csv
rm -rf $OUT/stats
# and the training leg on its own (forward(train) + backward + Adam), inference steps reduced to the minimum
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-extra-legs --train-precision bf16 --train-steps 10 > $OUT/${TAG}_bench_train_under_rocprof.json 2>> $OUT/bench.err
cp $OUT/stats/*/*kernel_stats.csv $OUT/${TAG}_rocprofv3_kernel_stats_train.csv
# ... and the matched-accuracy training mode (split-fp16 dgrad / wgrad kernels)
rm -rf $OUT/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --precision fp16x3 --train-precision fp16x3 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-extra-legs --train-steps 6 > $OUT/${TAG}_bench_train_fp16x3_under_rocprof.json 2>> $OUT/bench.err
cp $OUT/stats/*/*kernel_stats.csv $OUT/${TAG}_rocprofv3_kernel_stats_train_fp16x3.csv
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python $ROOT/bench.py --steps 2 --warmup 1 --no-kernel-timing $HEAD > /dev/null 2>> $OUT/bench.err
python $ROOT/tools/pmc_summary.py $OUT/pmc_sq > $OUT/${TAG}_pmc_sq_counters.txt
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_fetch_tr $OUT/pmc_write_tr
ls -la $OUT

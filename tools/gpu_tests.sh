cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu -x --durations=15 "$@" > gpurun_out/gpu_tests.log 2>&1; echo rc=$?
grep -v "^$" gpurun_out/gpu_tests.log | tail -45

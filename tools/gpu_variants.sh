cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/variants.py "$@" 2>&1 | tee gpurun_out/variants.log

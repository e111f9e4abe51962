#!/bin/bash
# Run ON the GPU box: the tests matching "$1" with full tracebacks
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$1" 2>&1 | tail -${2:-60}

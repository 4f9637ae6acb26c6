#!/bin/bash
# trilinear-geometry experiment: P=4 tests through a dev lib, then block-size sweep
export FUSMI_LIB=$PWD/abl/libfusmi_$1.so
timeout -k 10 600 python -m pytest tests/test_gpu_trilinear.py -x -q -m gpu -k "$2" > gpurun_out/tri_tests.log 2>&1
tail -5 gpurun_out/tri_tests.log
shift; shift
bash tools/gpu_sweep.sh "$@"

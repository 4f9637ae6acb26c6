#!/bin/bash
# usage: gpu_quick.sh [bench args]   -- GPU tests + short bench summary (run through gpurun)
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2> gpurun_out/gpu_tests.err; tail -3 gpurun_out/gpu_tests.log
timeout -k 10 300 python bench.py --no-cpu "$@" > gpurun_out/bench_quick.log 2> gpurun_out/bench_quick.err
tail -1 gpurun_out/bench_quick.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('value %.4e  ms/step %.3f  stiff_ms %.4f  kfrac %.3f  stepfrac %.3f'%(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['step_roofline']['frac_of_8TBps']))
print(d['kernel_ms_per_step'])" || tail -20 gpurun_out/bench_quick.log

#!/bin/bash
# usage: gpu_abl.sh "bench args"  -- product vs ablation builds (abl/*.so), same bench
for lib in "" abl/libfusmi_abl1.so abl/libfusmi_abl2.so; do
  if [ -n "$lib" ]; then export FUSMI_LIB=$PWD/$lib; else unset FUSMI_LIB; fi
  timeout -k 10 300 python bench.py --no-cpu --steps 10 --warmup 2 $1 > gpurun_out/bench_abl.log 2> gpurun_out/bench_abl.err
  tail -1 gpurun_out/bench_abl.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('lib=[$lib] $1: ms/step %.3f stiff_ms %.4f'%(d['ms_per_step'], d['roofline']['avg_launch_ms']))" || tail -5 gpurun_out/bench_abl.log
done

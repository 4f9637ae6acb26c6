#!/bin/bash
# usage: gpu_loopback.sh [lib name]  -- RK4 step with and without the looped-back RCCL exchange
[ -n "$1" ] && export FUSMI_LIB=$PWD/abl/libfusmi_$1.so
for a in "--both-geometries 0" "--halo-loopback" "--both-geometries 0 --no-profile" "--halo-loopback --no-profile"; do
  timeout -k 10 300 python bench.py --no-cpu --steps 20 --warmup 3 $a > gpurun_out/bench_lb.log 2> gpurun_out/bench_lb.err
  tail -1 gpurun_out/bench_lb.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('[$a] value %.4e ms/step %.3f stiff_ms %.4f'%(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']), {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})" || tail -8 gpurun_out/bench_lb.log
done

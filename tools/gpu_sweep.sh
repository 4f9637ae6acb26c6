#!/bin/bash
# usage: gpu_sweep.sh "args1" "args2" ...  -- one short bench per argument string
for a in "$@"; do
  timeout -k 10 300 python bench.py --no-cpu --steps 10 --warmup 2 $a > gpurun_out/bench_sweep.log 2> gpurun_out/bench_sweep.err
  tail -1 gpurun_out/bench_sweep.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$a: geom %s value %.4e ms/step %.3f stiff_ms %.4f kfrac %.3f stepfrac %.3f lds %d blocks %d'%(d['config']['geometry'][:7], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['step_roofline']['frac_of_8TBps'], d['config']['lds_bytes_per_block'], d['config']['blocks']))" || tail -5 gpurun_out/bench_sweep.log
done

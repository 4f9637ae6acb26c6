#!/bin/bash
# usage: gpu_variants_args.sh "name|bench args" ...  -- one bench per (experiment lib, argument string)
for spec in "$@"; do
  name="${spec%%|*}"; args="${spec#*|}"
  export FUSMI_LIB=$PWD/abl/libfusmi_$name.so
  timeout -k 10 300 python bench.py --no-cpu --steps 10 --warmup 2 --both-geometries 0 $args > gpurun_out/bench_var_$name.log 2> gpurun_out/bench_var_$name.err
  tail -1 gpurun_out/bench_var_$name.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('[$name $args] value %.4e ms/step %.3f stiff_ms %.4f kfrac %.3f stepfrac %.3f lds %d blocks %d'%(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['step_roofline']['frac_of_8TBps'], d['config']['lds_bytes_per_block'], d['config']['blocks']))" || tail -5 gpurun_out/bench_var_$name.log
done

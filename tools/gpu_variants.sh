#!/bin/bash
# usage: gpu_variants.sh "bench args" name1 name2 ...  -- same bench through abl/libfusmi_<name>.so
# (experiment builds made with `python fenicsx-fus_amd/build.py --dev --name <name> -D...`)
args="$1"; shift
for name in "$@"; do
  export FUSMI_LIB=$PWD/abl/libfusmi_$name.so
  timeout -k 10 300 python bench.py --no-cpu --steps 10 --warmup 2 $args > gpurun_out/bench_var_$name.log 2> gpurun_out/bench_var_$name.err
  tail -1 gpurun_out/bench_var_$name.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
o=d.get('other_geometry') or {}
print('[$name] value %.4e ms/step %.3f stiff_ms %.4f kfrac %.3f | other %.4e'%(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], o.get('value',0)))" || tail -5 gpurun_out/bench_var_$name.log
done

#!/bin/bash
# usage: gpu_diag_sweep.sh <out.txt>   (through gpurun) -- affine path on the box mesh, general form against the
# diagonal-metric form (option diag_metric), per degree and precision at 64^3
out=gpurun_out/$1
: > $out
for cfg in "--P 2" "--P 3" "--P 4" "--P 5" "--P 6" "--P 7" "--P 4 --dtype f32" "--P 5 --dtype f32" "--P 6 --dtype f32" "--P 7 --dtype f32" "--model lossy" "--model westervelt"; do
  for dm in 0 1; do
    timeout -k 10 400 python bench.py --no-cpu --traffic none --both-geometries 0 --repeats 3 --geometry auto --diag-metric $dm $cfg > gpurun_out/dsw_tmp.json 2>> gpurun_out/dsw.err \
      && python - "$cfg" $dm >> $out <<'PY'
import json,sys
d=json.load(open("gpurun_out/dsw_tmp.json"))
print("%-22s diag_metric=%s  %.4g DOF-upd/s  %.4f ms/step  kernel %.4f ms  pk=%s" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["config"]["packed_fp32"]))
PY
  done
done
cat $out

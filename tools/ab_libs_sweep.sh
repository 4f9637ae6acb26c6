#!/bin/bash
# usage: ab_libs_sweep.sh <out.txt> <libA.so> <libB.so> ...   (through gpurun)
# One bench line per library and configuration (degrees / precisions / geometry modes at 64^3), libraries
# interleaved per configuration: the per-configuration A/B of a full-library variant (compiler flags).
out=gpurun_out/$1; shift
: > $out
for cfg in "--P 3" "--P 4" "--P 4 --geometry auto" "--P 5" "--P 5 --geometry auto" "--P 6" "--P 6 --geometry auto" "--P 7" "--P 7 --geometry auto" \
           "--P 4 --dtype f32" "--P 5 --dtype f32" "--P 6 --dtype f32" "--P 7 --dtype f32" "--P 4 --geometry stream" "--P 7 --geometry stream" "--model lossy" "--P 8"; do
  for lib in "$@"; do
    FUSMI_LIB=$PWD/$lib timeout -k 10 400 python bench.py --no-cpu --traffic none --both-geometries 0 --repeats 3 $cfg > gpurun_out/abs_tmp.json 2>> gpurun_out/abs.err \
      && python - "$cfg" "$lib" >> $out <<'PY'
import json,sys
d=json.load(open("gpurun_out/abs_tmp.json"))
print("%-28s %-30s %.4g DOF-upd/s  %.4f ms/step  kernel %.4f ms" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
PY
  done
done
cat $out

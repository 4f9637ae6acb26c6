#!/bin/bash
# usage: ab_libs.sh <out-prefix> "<bench args>" <lib1.so> <lib2.so> ...   (through gpurun)
# A/B of library builds (FUSMI_LIB), round-robin over 3 rounds in separate processes.
out=$1; args=$2; shift 2
mkdir -p gpurun_out
for round in 1 2 3; do
  for lib in "$@"; do
    FUSMI_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu --traffic none --both-geometries 0 --repeats 3 $args > gpurun_out/${out}_tmp.json 2>> gpurun_out/${out}.err || { echo "$lib failed"; tail -3 gpurun_out/${out}.err; continue; }
    python - "$lib" $round <<PY
import json,sys
d=json.load(open("gpurun_out/${out}_tmp.json"))
print("round %s %-28s [$args]: %.4f ms/step, kernel %.4f ms, value %.4g" % (sys.argv[2], sys.argv[1], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["value"]))
PY
  done
done

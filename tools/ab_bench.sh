#!/bin/bash
# usage: ab_bench.sh <out-prefix> "<common bench args>" "<variant args 1>" "<variant args 2>" ...   (through gpurun)
# Runs the variants round-robin (3 rounds) in separate processes and prints ms_per_step / kernel time per run.
out=$1; common=$2; shift 2
mkdir -p gpurun_out
for round in 1 2 3; do
  i=0
  for v in "$@"; do
    i=$((i+1))
    timeout -k 10 300 python bench.py --no-cpu --traffic none --both-geometries 0 --repeats 3 $common $v > gpurun_out/${out}_v${i}_r${round}.json 2>> gpurun_out/${out}.err || { echo "variant $i failed"; tail -3 gpurun_out/${out}.err; }
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${out}_v${i}_r${round}.json"))
    print("round ${round} variant ${i} [$v]: %.4f ms/step, kernel %.4f ms, value %.4g, stage %.3f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["value"], d["kernel_ms_per_step"].get("stage",0)))
except Exception as e:
    print("round ${round} variant ${i}: no result", e)
PY
  done
done

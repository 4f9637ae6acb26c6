#!/bin/bash
# usage: ab_matrix.sh <out-prefix> <rounds> "<common bench args>" -- <lib1.so> <lib2.so> ... -- "<variant args 1>" "<variant args 2>" ...
# (through gpurun)  Every (library build, variant) pair, round-robin over <rounds> rounds in separate processes;
# one line per run: ms per step, block-kernel ms per launch, DOF-updates/s.
out=$1; rounds=$2; common=$3; shift 4
libs=(); while [ "$1" != "--" ]; do libs+=("$1"); shift; done; shift
mkdir -p gpurun_out
: > gpurun_out/${out}.txt
for round in $(seq 1 $rounds); do
  for v in "$@"; do
    for lib in "${libs[@]}"; do
      FUSMI_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu --traffic none --both-geometries 0 --repeats 3 --steps 20 $common $v > gpurun_out/${out}_tmp.json 2>> gpurun_out/${out}.err || { echo "$lib [$v] failed" | tee -a gpurun_out/${out}.txt; tail -3 gpurun_out/${out}.err; continue; }
      python - "$lib" $round "$v" <<PY | tee -a gpurun_out/${out}.txt
import json,sys
d=json.load(open("gpurun_out/${out}_tmp.json"))
print("round %s %-26s [%s]: %.4f ms/step, kernel %.4f ms, stage %.3f, value %.4g, blocks %d lds %d" % (sys.argv[2], sys.argv[1].split("/")[-1], sys.argv[3], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["kernel_ms_per_step"].get("stage",0), d["value"], d["config"]["blocks"], d["config"]["lds_bytes_per_block"]))
PY
    done
  done
done

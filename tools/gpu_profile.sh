#!/bin/bash
# usage: gpu_profile.sh <tag> ["extra bench args"]   (through gpurun) -- the evidence behind bench.py's roofline object:
#   1. rocprofv3 --kernel-trace --stats of the bench command (the program itself directly after `--`)
#   2. the un-profiled bench line of the same command (bench.py measures the HBM traffic of the dominant kernel
#      itself: two child passes under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, see bench.py live_traffic)
# Outputs in gpurun_out/prof_<tag>/: <tag>_kernel_stats.csv, <tag>_bench.json (copy to profiles/).
export TMPDIR=/tmp
tag=$1
extra="$2"
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu --traffic none --steps 20 --warmup 3 --repeats 1 --both-geometries 0 $extra > $out/bench_under_rocprof.log 2>&1 || { tail -5 $out/bench_under_rocprof.log; exit 1; }
echo "stats pass done"
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
timeout -k 10 1100 python bench.py --steps 20 --warmup 3 $extra > $out/${tag}_bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python - <<PY
import json
d=json.load(open("$out/${tag}_bench.json"))
r=d["roofline"]
print("$tag: %.4g DOF-updates/s, %.4f ms/step, kernel %.4f ms, frac %.3f, frac_real %s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["frac_real"]))
PY

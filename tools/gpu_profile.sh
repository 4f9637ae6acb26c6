#!/bin/bash
# usage: gpu_profile.sh <tag> ["extra bench args"]   (through gpurun) -- the evidence behind bench.py's roofline object:
#   1. rocprofv3 --kernel-trace --stats of the default bench command
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of a short run of the same workload
#   3. the un-profiled bench line
# then tools/make_profile_summary.py condenses them into gpurun_out/prof_<tag>/summary/ (copy to profiles/).
export TMPDIR=/tmp
tag=$1
extra="$2"
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 20 --warmup 3 $extra > $out/bench_under_rocprof.log 2>&1 || { tail -5 $out/bench_under_rocprof.log; exit 1; }
echo "stats pass done"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$ctr -- python3 bench.py --no-cpu --steps 4 --warmup 1 --both-geometries 0 $extra > $out/pmc_$ctr.log 2>&1 || { tail -5 $out/pmc_$ctr.log; exit 1; }
  echo "pmc $ctr done"
done
timeout -k 10 600 python bench.py --steps 20 --warmup 3 $extra > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python tools/make_profile_summary.py $tag "$extra"

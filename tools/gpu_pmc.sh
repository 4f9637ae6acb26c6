#!/bin/bash
# usage: gpu_pmc.sh <tag> "<counters>" "<bench args>" -- one rocprofv3 --pmc pass, summarised per kernel
export TMPDIR=/tmp
tag=$1; ctr=$2; args=$3
rm -rf gpurun_out/pmc_$tag
timeout -k 10 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --no-cpu --steps 4 --warmup 1 $args > gpurun_out/pmc_$tag.log 2>&1
f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
f=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
seen=set()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0][:48]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    key=(k,r["Dispatch_Id"])
    if key not in seen: seen.add(key); cnt[k]+=1
for k in acc:
    if cnt[k]<4: continue
    print(k, "dispatches", cnt[k], {c: "%.4g"%(v/cnt[k]) for c,v in acc[k].items()})
PY

#!/bin/bash
# usage: gpu_sweep.sh <out.txt>   (through gpurun) -- one bench line per degree / precision / model at 64^3
out=gpurun_out/$1
: > $out
run() {
  timeout -k 10 400 python bench.py --no-cpu --traffic none --both-geometries 0 --repeats 3 $@ > gpurun_out/sweep_tmp.json 2>> gpurun_out/sweep.err \
    && python - "$*" >> $out <<'PY'
import json,sys
d=json.load(open("gpurun_out/sweep_tmp.json"))
r=d["roofline"]
print("%-46s %.4g DOF-upd/s  %.4f ms/step  kernel %.4f ms  frac(8d) %.3f  frac_compulsory %.3f  step frac_compulsory %.3f  pk=%s blocks=%d lds=%d" % (sys.argv[1], d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["frac_compulsory"], d["step_roofline"]["frac_compulsory"], d["config"]["packed_fp32"], d["config"]["blocks"], d["config"]["lds_bytes_per_block"]))
PY
}
for P in 2 3 4 5 6 7; do run --P $P; run --P $P --geometry auto; done
for P in 3 4 5 6 7; do run --P $P --dtype f32; run --P $P --dtype f32 --geometry auto; done
run --P 8; run --P 8 --geometry auto; run --P 8 --dtype f32; run --P 9 --cells 32; run --P 10 --cells 32; run --P 8 --cells 32 --geometry stream
run --model lossy; run --model westervelt
run --P 4 --geometry stream; run --P 7 --geometry stream
cat $out

"""Condenses the rocprofv3 outputs of tools/gpu_profile.sh into the three files kept under profiles/:
<tag>_kernel_stats.csv (rocprofv3 --stats table), <tag>_pmc_traffic.json (HBM bytes per launch from
the FETCH_SIZE / WRITE_SIZE passes with the gfx950 correction) and <tag>_bench.json."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
extra = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.join("gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "summary")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(root, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))


def per_kernel(counter):
    f = glob.glob(os.path.join(root, f"pmc_{counter}", "**", "*counter_collection.csv"), recursive=True)[0]
    acc, seen = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0]
        acc[k] += float(r["Counter_Value"])
        seen[k].add(r["Dispatch_Id"])
    return {k: (acc[k] / len(seen[k]), len(seen[k])) for k in acc}


fetch, write = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
per = {}
for k in sorted(set(fetch) | set(write)):
    per[k] = {"dispatches": fetch.get(k, write.get(k))[1], "FETCH_SIZE_per_dispatch": fetch.get(k, (0, 0))[0],
              "WRITE_SIZE_per_dispatch": write.get(k, (0, 0))[0]}
# the fused stiffness + RK-stage variants of the block kernel: <T, P, OP=0, ATOMIC, STAGE in {0,1,3}, NF, GEOM, TD>
fused = [k for k in per if "k_block_op<double, 4, 0, 1, " in k and not k.split("<")[1].startswith("double, 4, 0, 1, -1")]
n = sum(per[k]["dispatches"] for k in fused)
rd = sum(per[k]["FETCH_SIZE_per_dispatch"] * per[k]["dispatches"] for k in fused) / max(n, 1) * 1024 * 2
wr = sum(per[k]["WRITE_SIZE_per_dispatch"] * per[k]["dispatches"] for k in fused) / max(n, 1) * 1024
out = {
    "command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE> --kernel-trace -- python3 bench.py --no-cpu --steps 4 --warmup 1 "
               "--both-geometries 0 " + extra + " (separate passes; tools/gpu_profile.sh)",
    "tag": tag,
    "config": "64^3 hex p=4 fp64, " + ("geometry per bench args: " + extra if extra else "general geometry (G streamed)")
              + ", default blocks, LDS atomics, fused RK4 stage epilogue",
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM); units KiB",
    "k_block_op_fused": {"kernels": fused, "launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                         "hbm_bytes_per_launch": rd + wr},
    "per_kernel": per,
}
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print("fused block kernel: %.3f GB read + %.3f GB written per launch over %d launches" % (rd / 1e9, wr / 1e9, n))

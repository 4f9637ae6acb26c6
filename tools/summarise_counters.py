"""Condenses the rocprofv3 --pmc passes of tools/gpu_counters.sh into one JSON per tag:
per kernel (dispatch count >= 4) the average counter values per dispatch, the average dispatch
duration from the kernel trace of the same passes, and a few ratios that say what the kernel waits on.

SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_LDS_IDX_ACTIVE and
SQ_LDS_BANK_CONFLICT count LDS-array cycles; SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE are summed over the 8
XCDs (MI355X_MICROARCH.md, rocprofv3 PMC slots / cycle constants)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
args = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.join("gpurun_out", f"ctr_{tag}")


def short(name):
    return name.split("(")[0].replace("fus::", "").replace("void ", "")


acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
dur = collections.defaultdict(list)
for p in sorted(glob.glob(os.path.join(root, "pass*"))):
    if not os.path.isdir(p):
        continue
    for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    for f in glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)

out = {"tag": tag, "command": "rocprofv3 --pmc <pass> --kernel-trace -- python3 bench.py --no-cpu --steps 4 --warmup 1 "
                              "--repeats 1 --both-geometries 0 " + args + " (three separate passes, tools/gpu_counters.sh)",
       "units": "counter values are averages per dispatch; us = average dispatch duration under the counter passes",
       "kernels": {}}
for k in acc:
    n = max(len(s) for s in disp[k].values())
    if n < 4:
        continue
    c = {name: acc[k][name] / len(disp[k][name]) for name in acc[k]}
    d = {"dispatches": n, "us": sum(dur[k]) / max(len(dur[k]), 1), "counters": c}
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    r = {}
    if wc > 0:
        for name in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if name in c:
                r[name + "/SQ_WAVE_CYCLES"] = c[name] / wc
    if c.get("SQ_LDS_IDX_ACTIVE", 0) > 0 and "SQ_LDS_BANK_CONFLICT" in c:
        r["SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if c.get("SQ_BUSY_CU_CYCLES", 0) > 0 and "SQ_LDS_IDX_ACTIVE" in c:
        r["SQ_LDS_IDX_ACTIVE/SQ_BUSY_CU_CYCLES"] = c["SQ_LDS_IDX_ACTIVE"] / c["SQ_BUSY_CU_CYCLES"]
    d["ratios"] = r
    out["kernels"][k] = d
dst = os.path.join(root, f"{tag}_counters.json")
json.dump(out, open(dst, "w"), indent=1)
for k, d in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["us"] * kv[1]["dispatches"])[:6]:
    print(k[:70], "n=%d %.1f us" % (d["dispatches"], d["us"]), {a: "%.3g" % b for a, b in d["ratios"].items()})
print("->", dst)

"""bench.py's output contract (one JSON line on stdout with the driver's keys plus `roofline` and
`cpu_baseline`), checked on a small instance of the workload, and its behaviour without a GPU (the
product has no CPU path: it must stop with an error, not print a number)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, timeout=600):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                          timeout=timeout, cwd=ROOT)


def test_bench_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("device present")
    out = run_bench("--steps", "1", "--warmup", "0", "--cells", "4", "--no-cpu")
    assert out.returncode != 0
    assert out.stdout.strip() == ""            # no result line


@pytest.mark.gpu
def test_bench_json_contract():
    out = run_bench("--steps", "3", "--warmup", "1", "--repeats", "3", "--cells", "16", "--cpu-n", "8", "--cpu-steps", "5",
                    "--traffic", "profile")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1                      # exactly one line on stdout, everything else on stderr
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str),
                     ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["unit"] == "DOF-updates/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    assert "model" not in d["config"]
    ndofs = (16 * 4 + 1) ** 3
    assert abs(d["value"] - ndofs * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert r["traffic"] is None and r["frac_real"] is None   # no committed PMC passes for this small configuration
    k2 = d["second_kernel"]                                   # the shared-dof stage kernel beside the dominant one
    assert k2["shared_dofs"] > 0 and k2["partial_sums"] >= 2 * k2["shared_dofs"] and 0 < k2["share_of_step"] < 1
    rp = d["repeats"]
    assert rp["n"] == 3 and len(rp["ms_per_step"]) == 3 and rp["min"] <= d["ms_per_step"] <= rp["max"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "DOF-updates/s" and c["sample"]
    # the other geometry paths are timed beside the headline at N = 1
    assert d["other_geometry"]["value"] > 0 and d["streamed_geometry"]["value"] > 0
    assert "trilinear" in d["config"]["geometry"]

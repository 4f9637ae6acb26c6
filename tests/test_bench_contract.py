"""bench.py's output contract (one JSON line on stdout with the driver's keys plus `roofline` and
`cpu_baseline`), checked on a small instance of the workload, and its behaviour without a GPU (the
product has no CPU path: it must stop with an error, not print a number)."""
import json
import os
import subprocess
import sys

import numpy as np
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


def test_self_launch_relays_rank0_line_and_exit_code(capsys, monkeypatch):
    """`python bench.py --gpus N` as a plain command (the driver's SCALE command): the parent starts the ranks as a child
    torch.distributed.run (never exec), relays rank 0's one JSON line and returns the children's exit code."""
    sys.path.insert(0, ROOT)
    import bench

    cmd = bench.launcher_command(8, ["--gpus", "8", "--steps", "5"], 29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5"]

    seen = {}

    class R:
        def __init__(self, rc, out):
            self.returncode, self.stdout = rc, out

    def fake_ok(cmd, stdout=None, env=None, cwd=None):
        seen["cmd"], seen["env"] = cmd, env
        return R(0, b'RCCL banner\n{"metric": "m", "n_gpus": 8}\n')

    monkeypatch.delenv("FUSMI_BENCH_REHEARSAL", raising=False)
    assert bench.self_launch(8, ["--gpus", "8"], ndev=8, run=fake_ok) == 0
    out = capsys.readouterr().out
    assert out == '{"metric": "m", "n_gpus": 8}\n'               # one line, nothing else
    assert "FUSMI_BENCH_REHEARSAL" not in seen["env"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["env"]["MASTER_ADDR"] == "127.0.0.1"
    assert seen["cmd"][seen["cmd"].index("--nproc-per-node=8")] and "exec" not in " ".join(seen["cmd"])

    # fewer devices than ranks: the ranks are told to share them (rehearsal)
    assert bench.self_launch(2, ["--gpus", "2"], ndev=1, run=fake_ok) == 0
    assert seen["env"]["FUSMI_BENCH_REHEARSAL"] == "1"
    capsys.readouterr()

    # a failing rank: non-zero exit code, no result line
    assert bench.self_launch(2, ["--gpus", "2"], ndev=2, run=lambda *a, **k: R(3, b"noise\n")) == 3
    assert capsys.readouterr().out == ""
    # ranks that exit 0 without printing a result are an error too
    assert bench.self_launch(2, ["--gpus", "2"], ndev=2, run=lambda *a, **k: R(0, b"")) != 0


def test_bench_gpus_2_without_gpu_fails_cleanly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("device present")
    out = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--cells", "4", "--no-cpu")
    assert out.returncode != 0 and out.stdout.strip() == ""
    assert "no HIP device" in out.stderr


@pytest.mark.gpu
def test_bench_two_ranks_self_launched_on_one_box():
    """The N > 1 path end to end through `python bench.py --gpus 2` (self-launch): two distinct rank processes, each
    with its own x-slab, exchanging their interface plane every stage.  On a one-GPU box the two ranks share the
    device and the exchange is staged over gloo (rehearsal); on a multi-GPU box it is the library's RCCL path."""
    out = run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "2", "--cells", "8", "--no-cpu",
                    "--traffic", "none", timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["finite_nonzero_solution"] is True
    assert d["comm"]["nranks"] == 2
    assert d["config"]["ndofs_global"] == (2 * 8 * 4 + 1) * (8 * 4 + 1) ** 2
    if d["comm"]["devices_visible"] < 2:
        assert "rehearsal" in d and d["comm"]["transport"] == "torch"
    else:
        assert d["comm"]["transport"] == "rccl" and "exchange_overlap" in d


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


def test_compulsory_bytes_model_and_slab_workload():
    """bench.py's compulsory-bytes roofline (the bytes THIS layout must move) on hand-computable layout statistics, and the
    --cells-xyz workload (one rank's x-slab of the 256^3 configs as a box of its own) -- host logic only."""
    sys.path.insert(0, ROOT)
    import argparse

    import bench

    info = {"interior_dofs": 1000, "shared_dofs": 200, "pairs": 450, "shapes": 2, "nblocks": 10}
    nc, N3, s = 80, 125, 8
    blk, sh, det = bench.compulsory_bytes(info, nc, N3, s, "trilinear", "linear", lean=True)
    assert det["x_block_local"] == (1000 + 450) * 8 and det["gather_index_and_partial_position"] == 8 * 450
    assert det["partial_sums_written"] == 450 * 8 and det["geometry_per_cell"] == 21 * 8 * 80
    assert det["stage_update_streams_interior"] == (4 + 5 + 6 + 7) / 4 * 8 * 1000          # lean RK4 stage kinds 4, 5, 6, 7
    assert abs(blk - sum(det.values())) < 1e-9 and sh == (450 + (5 + 6 + 7 + 8) / 4 * 200) * 8
    blk_s, _, det_s = bench.compulsory_bytes(info, nc, N3, s, "stream", "linear", lean=False)
    assert det_s["geometry_per_cell"] == 6 * N3 * 8 * 80 and det_s["stage_update_streams_interior"] == (7 + 10 + 10 + 6) / 4 * 8 * 1000
    assert blk_s > blk
    # two gathered operator inputs (Lossy), two more vector reads (Westervelt)
    _, _, det_l = bench.compulsory_bytes(info, nc, N3, s, "affine", "lossy")
    assert det_l["x_block_local"] == 2 * (1000 + 450) * 8 and det_l["geometry_per_cell"] == 7 * 8 * 80
    args = argparse.Namespace(P=2, global_cells=0, cells=4, cells_xyz=(2, 4, 3), medium="skull")
    mesh, V, tags, c, rho, freq, p0, dt = bench.workload(args, 0, 1)
    assert mesh.num_cells == 24 and V.num_dofs == 5 * 9 * 7 and len(c) == 24
    h = 0.12 / 64
    assert abs(dt - (1 / freq) / np.ceil((1 / freq) / (0.5 * h / (2800.0 * 4)))) < 1e-18
    mesh2, V2, *_ = bench.workload(args, 1, 2)                  # weak scaling: the same box per rank
    assert mesh2.num_cells == 24 and V2.global_offset == 2 * 2 * 9 * 7

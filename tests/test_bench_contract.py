"""bench.py prints one JSON line with the contract's keys (a small field so that it runs in seconds)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--height", "300", "--width", "260",
                          "--sources", "60", "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "sources/sec"
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["scaling"] == "strong"
    assert d["config"]["shard_sizes"] == [60] and d["host_api_sources_per_sec"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    # the binding roofline on top (the fused kernel is FP64-vector-ALU bound: SURVEY.md 8(d)); the HBM figures nested
    assert r["bound"] == "fp64_valu" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    h = r["hbm"]
    assert h["bound"] == "hbm" and h["unit"] == "GB/s" and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-12 and "traffic" in h
    assert d["parity_pin"] in ("absent", "present") and "timing_method" in d
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["value"] > 0 and d["split_variant"]["kernel"] == "record_sum_kernel" and d["optimizer"]["failed"] == 0
    # what the engine does where the reference calls it: one elbo() per call, a Cyclades-sized layer, the
    # joint-inference schedule through celeste_joint_infer, rank 0's shard for N ranks
    assert d["single_call_latency_us"]["median"] > 0 and d["single_call_latency_us"]["calls"] == 200
    assert d["single_call_latency_us"]["per_source_context"]["median"] > 0
    lay = d["optimizer"]["cyclades_layer"]
    assert lay["failed"] == 0 and lay["us_per_newton_iteration_of_the_slowest_target"] > 0 and lay["driver"] == "fused"
    assert d["joint_infer"]["seconds"] > 0 and d["joint_infer"]["layers"] >= 3 and d["joint_infer"]["failed"] == 0
    sp = d["shard_projection"]
    assert sp["1"]["targets_rank0"] == 60 and sp["8"]["targets_rank0"] < sp["2"]["targets_rank0"] and sp["8"]["ms_per_sweep"] > 0
    v = r["valu"]
    assert v["pixel_visits"] == d["config"]["pixel_visits_per_sweep"] and 0 < v["frac"] < 1 and v["instruction_mix"]["fma_class"] > 0
    assert "config5" not in d          # (appended to full-size config-3 runs only: it generates 16 SDSS-size fields)
    assert "variable_psf" not in d     # (likewise: a full-size field with one PSF stamp per patch)
    # where the step goes: the timed loop's steps one by one (HIP events), what no kernel accounts for, the clock behind the loop
    sm = d["step_ms"]
    assert 0 < sm["min"] <= sm["p50"] <= sm["max"] and sm["sum"] <= sm["wall_ms"] * 1.05
    assert abs(d["gaps_ms"] - (d["ms_per_step"] - sum(d["kernels_ms"].values()))) < 1e-9
    assert "sclk_mhz_after_loop" in d and (d["sclk_mhz_after_loop"] is None or d["sclk_mhz_after_loop"] > 50)
    assert d["steady_state"]["steps"] == 200 and d["steady_state"]["ms_per_step"] > 0      # (the timed loop was short)
    # ConstantPSFMap: one stamp per band (two bands of the synthetic recipe share a PSF width, hence a stamp)
    assert h["psf_stamps"] in (4, 5) and h["spline_coefficient_bytes"] == h["psf_stamps"] * 53 * 53 * 8


@pytest.mark.gpu
def test_bench_variable_psf_flag():
    """--variable-psf: the same sweep on a field whose PSF is an SDSSPSFMap evaluated at every source -- one stamp per patch, its
    spline coefficients counted per patch in the algorithmic bytes"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--height", "300", "--width", "260", "--sources", "60",
                          "--steps", "3", "--warmup", "1", "--variable-psf", "--no-extras", "--no-cpu-baseline", "--no-live-pmc"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    h = d["roofline"]["hbm"]
    assert "SDSSPSFMap" in d["config"]["workload"] and 200 <= h["psf_stamps"] <= 300
    assert h["spline_coefficient_bytes"] == h["psf_stamps"] * 53 * 53 * 8 < h["algorithmic_bytes_per_launch"]
    assert d["value"] > 0


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "no CPU fallback" in (out.stderr + out.stdout)


def test_bench_refuses_a_rank_count_that_is_not_what_gpus_says():
    """started by a launcher with fewer (or more) ranks than --gpus: no line, non-zero exit -- never a line that says
    n_gpus 1 for a run that was asked to be 8 (checked before anything touches a device, so it runs here)"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True,
                         timeout=600, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr and "{" not in out.stdout
    env["WORLD_SIZE"] = "2"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr and "{" not in out.stdout


def test_bench_plain_invocation_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-runs itself under torch.distributed.run with N processes on
    127.0.0.1 (the command is checked here; the ranks themselves need a GPU: test_gpu_round2.py)"""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    import subprocess as sp
    monkeypatch.setattr(sp, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    assert bench.self_launch(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    # --standalone: the launcher binds a free rendezvous port itself; on 127.0.0.1 (the hostname may not resolve)
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    assert cmd[-5].endswith("bench.py") and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


@pytest.mark.gpu
def test_bench_config5_mode_small_grid():
    """bench.py --config 5: overlapping fields through the sparse patch list, fp32 component loop, the 1e-4 check
    against the fp64 device path inside the run (a 2 x 2 grid of small fields so that it runs in seconds)"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "5", "--grid", "2,2", "--height", "260",
                          "--width", "240", "--sources", "150", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert d["dtype"] == "f32" and "configs[4]" in d["config"]["workload"] and d["config"]["shard_sizes"] == [150]
    chk = d["fp32_vs_fp64_device"]
    assert chk["sources_checked"] == 150 and max(chk["v"], chk["d"], chk["h"]) <= 1e-4
    assert d["roofline"]["kernel"] == "pixel_kernel<2, float>" and d["value"] > 0

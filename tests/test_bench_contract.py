"""bench.py pieces that run without a GPU: the FLOP model of SURVEY.md 8(d), the roofline blocks
(fields the driver's contract names) and the CPU-baseline port used for `cpu_baseline`."""
import argparse
import importlib.util
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_flop_model_matches_survey():
    b = _bench()
    assert sum(b.flops_per_pose(2, 1024)) == 25591808          # SURVEY.md 8(d), BASELINE configs[1]
    assert b.flops_per_pose(2, 1024) == (8552448, 17039360)
    assert sum(b.flops_per_pose(4, 1024)) == 50757632
    assert sum(b.flops_per_pose(8, 2048)) == 403505152


@pytest.mark.parametrize("dtype,bound,peak", [("fp32", "mfma", 157.3), ("bf16x3", "mfma", 2500.0 / 6),
                                              ("fp16x2", "mfma", 2500.0 / 3), ("bf16s", "mfma", 2500.0)])
def test_roofline_block_fields(dtype, bound, peak):
    b = _bench()
    args = argparse.Namespace(batch=4096, width=1024, dtype=dtype)
    dom = {"ms": 0.072, "tflops": 2.0 * 4096 * 1024 * 1024 / 0.072 / 1e9}
    r = b.roofline_block(args, dom)
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r
    assert r["bound"] == bound and abs(r["peak"] - peak) < 1e-6 * peak
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    if dtype == "fp32":          # PMC traffic of the dominant kernel, recorded under profiles/
        import json, os
        rec = json.load(open(os.path.join(os.path.dirname(__file__), "..", "profiles", "r04_traffic.json")))
        k = [v for n, v in rec["shapes"]["4096x1024"].items() if n.startswith("gemm_f32_ring_kernel<128, 128, 4, 2, 0, 0, 2")][0]
        assert k["traffic_bytes"] == 2 * 1024 * k["fetch_size_kb"] + 1024 * k["write_size_kb"]   # gfx950 x2 correction
        assert r["traffic"] == k["traffic_bytes"] and abs(r["achieved"] - 119.3) < 0.1
        assert "r04_traffic.json" in r["traffic_unit"]          # this round's binary, not an older record


def test_cpu_baseline_port_runs_the_whole_step():
    from oracle import torch_port as TP
    out = TP.time_cpu_steps(1, 64, 32, steps=2, warmup=1, threads=2)
    assert out["steps"] == 2 and out["threads"] == 2
    assert out["poses_per_s"] > 0 and abs(out["ms_per_step"] * out["poses_per_s"] / 1e3 - 32) < 1e-6 * 32


def test_pre_ramp_leaves_on_the_agreed_clock(monkeypatch):
    """N > 1: every rank must leave the pre-ramp after the same number of steps (each step holds collectives),
    so the exit test uses the value `agree` returns (the maximum over ranks), never the local clock alone."""
    import torch
    b = _bench()
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    steps = []
    answers = iter([0.0, 0.0, 1e9])          # the group says "not yet" twice, whatever this rank's clock shows
    el, n = b.pre_ramp(lambda: steps.append(1), 1e-6, agree=lambda ms: next(answers))
    assert n == 30 and len(steps) == 30 and el == 1e9
    steps.clear()
    el, n = b.pre_ramp(lambda: steps.append(1), 1e-6)       # one rank: its own clock
    assert n == 10 and len(steps) == 10

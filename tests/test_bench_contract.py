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
        rec = json.load(open(os.path.join(os.path.dirname(__file__), "..", "profiles", "r06_hbm_traffic.json")))
        assert "commit" in rec["code"]                          # the record names the code it was taken on
        cfg = rec["configs"]["configs[1]"]
        assert cfg["shape"]["B"] == 4096 and cfg["shape"]["W"] == 1024
        k = [v for n, v in cfg["kernels"].items() if n.startswith("gemm_f32_ring_kernel<128, 128, 4, 2, 0, 0, 2")][0]
        assert k["traffic_bytes"] == k["read_bytes"] + k["write_bytes"]     # read side = 2 * FETCH_SIZE (gfx950)
        assert r["traffic"] == k["traffic_bytes"] and abs(r["achieved"] - 119.3) < 0.1
        assert "r06_hbm_traffic.json" in r["traffic_unit"]      # this round's binary, not an older record
        # the same kernel's round-4 record (unchanged source since): within 0.1 %
        old = json.load(open(os.path.join(os.path.dirname(__file__), "..", "profiles", "r04_traffic.json")))
        k4 = [v for n, v in old["shapes"]["4096x1024"].items() if n.startswith("gemm_f32_ring_kernel<128, 128, 4, 2, 0, 0, 2")][0]
        assert abs(k4["traffic_bytes"] - k["traffic_bytes"]) < 1e-3 * k["traffic_bytes"]
    if dtype == "bf16s":
        assert r["traffic"] is None or "r0" in r["traffic_unit"]


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


# ----------------------------------------------------------------------------------------------------------------
# N > 1 lines must prove what they ran on: the ``comm`` block, distinct devices, rehearsal labels
def _ident(host, pci, index=0, uuid=None):
    return {"hostname": host, "pid": 1, "device_index": index, "visible": None, "pci_bus_id": pci, "uuid": uuid,
            "name": "AMD Instinct MI355X"}


def test_n_gpus_counts_distinct_devices_not_ranks():
    b = _bench()
    eight = [_ident("node", "0000:%02x:00.0" % (5 + 16 * i), i) for i in range(8)]
    assert b.distinct_devices(eight) == 8
    assert b.distinct_devices([_ident("node", "0000:05:00.0")] * 8) == 1           # 8 ranks time-slicing one GPU
    assert b.distinct_devices([_ident("a", "0000:05:00.0"), _ident("b", "0000:05:00.0")]) == 2    # two hosts
    assert b.distinct_devices([_ident("a", None, 0, "GPU-1"), _ident("a", None, 1, "GPU-1")]) == 1   # UUID fallback
    assert b.distinct_devices([_ident("a", None, 0), _ident("a", None, 1)]) == 2   # index fallback


def test_workload_label_never_claims_gpus_it_did_not_hold():
    b = _bench()
    a = argparse.Namespace(blocks=2, width=1024, batch=4096, dtype="fp32")
    assert b.workload_label(a, 1).startswith("BASELINE configs[1]") and "run on" not in b.workload_label(a, 1)
    reh = "8 ranks on 1 GPU over gloo: control flow only, NOT a multi-GPU measurement"
    lab = b.workload_label(a, 1, reh)
    assert "REHEARSAL" in lab and "run on 1 GPU" in lab and "run on 8 GPUs" not in lab
    a3 = argparse.Namespace(blocks=4, width=1024, batch=8192, dtype="bf16s")
    assert "run on" not in b.workload_label(a3, 8)                 # configs[3] really on 8 GPUs
    assert "run on 2 GPUs" in b.workload_label(a3, 2)
    assert "REHEARSAL" in b.workload_label(a3, 8, reh)             # even with a matching count a rehearsal says so


def _comm_worker(rank, world, port, out_dir, rehearse):
    import json
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = _bench()
        # rehearsal: both ranks hold the same device; otherwise one device each
        ident = _ident("box", "0000:05:00.0" if rehearse else "0000:%02x:00.0" % (5 + rank), 0 if rehearse else rank)
        comm = b.comm_block(ident, rehearse)
        with open(os.path.join(out_dir, "comm%d.json" % rank), "w") as f:
            json.dump(comm, f)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rehearse", [False, True])
def test_comm_block_world2_gloo(tmp_path, rehearse):
    """Every rank ends with the same record: backend and world size as the group reports them, every rank's device,
    the number of DISTINCT devices; a rehearsal is labelled as such."""
    import json
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    mp.spawn(_comm_worker, args=(2, port, str(tmp_path), rehearse), nprocs=2, join=True)
    c0, c1 = (json.load(open(tmp_path / ("comm%d.json" % r))) for r in (0, 1))
    assert c0 == c1
    assert c0["backend"] == "gloo" and c0["world_size"] == 2
    assert [r["rank"] for r in c0["ranks"]] == [0, 1]
    for key in ("hostname", "pid", "device_index", "pci_bus_id", "uuid", "name"):
        assert all(key in r for r in c0["ranks"])
    assert c0["n_distinct_devices"] == (1 if rehearse else 2) and c0["n_hosts"] == 1
    assert "rccl_version_torch_built_with" in c0 and "rccl_version_loaded" in c0
    if rehearse:
        assert c0["rehearsal"].startswith("2 ranks on 1 GPU over gloo")
    else:
        assert c0["rehearsal"] is None


def test_bench_source_reports_distinct_devices_and_nulls_a_rehearsal_value():
    """The line's n_gpus is comm.n_distinct_devices (never WORLD_SIZE) and a rehearsal carries no `value`."""
    src = open(os.path.join(REPO, "bench.py")).read()
    assert '"n_gpus": n_devices' in src and '"n_gpus": world' not in src
    assert '"value": None if rehearsal else poses' in src
    assert "os.exec" not in src and "execv" not in src             # a process that touched the GPU is never replaced


def test_reap_process_group_leaves_no_children(tmp_path):
    """self_launch starts the launcher in a session of its own; whatever is still alive in that group when the
    launcher has exited is terminated (a rank stuck in a collective after another rank raised)."""
    import subprocess
    import sys
    import time
    b = _bench()
    # a "launcher" that starts a sleeping "rank" and exits at once with a failure code
    code = ("import subprocess, sys; subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)']); "
            "sys.exit(3)")
    proc = subprocess.Popen([sys.executable, "-c", code], start_new_session=True)
    assert proc.wait() == 3
    time.sleep(0.2)
    assert b.reap_process_group(proc.pid, grace_s=3.0) == 1        # the orphaned sleeper was found and ended
    assert b.reap_process_group(proc.pid, grace_s=0.5) == 0        # nothing left

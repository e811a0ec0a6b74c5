#!/usr/bin/env python3
"""Benchmark of the lifter training step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py
     --gpus N ...,  or bare: bench.py then starts the N ranks itself, before any GPU call)

A "step" is the whole step body of /root/reference/train_bilinear.py:75-83 on one
synthetic batch resident in HBM: zero_grad, forward, MSELoss, backward,
[gradient all-reduce when N > 1], clip_grad_norm_(1), Adam.step.  The workload at N=1
is BASELINE.json configs[1]: 2 residual blocks, width 1024, batch 4096, fp32
(weak scaling: every GPU processes its own 4096-pose batch).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     : achieved fraction of the fp32-MFMA peak of the dominant kernel
                 (the 1024x1024 Linear forward GEMM), timed live with HIP events;
  cpu_baseline : oracle/torch_port.py (plain PyTorch on the host cores) on the same
                 workload, bounded sample, rank 0 at N = 1 only;
  configs      : (default N = 1 run) the other single-GPU BASELINE.json shapes — configs[2] and the
                 per-GPU shapes of configs[3] / configs[4], bf16 storage — each with its own
                 ms_per_step / value / roofline, so that one driver run shows every config.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, Peak FP32 (matrix)
BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA
HBM_PEAK_GBS = 8000.0


def flops_per_pose(num_blocks, width):
    """fwd = 2*32*W + L*2*W^2 + 2*W*48; bwd = 2*32*W + L*4*W^2 + 4*W*48."""
    L = 2 * num_blocks
    fwd = 2 * 32 * width + L * 2 * width * width + 2 * width * 48
    bwd = 2 * 32 * width + L * 4 * width * width + 4 * width * 48
    return fwd, bwd


def time_kernel(fn, reps, warm=50):
    """Average duration (ms) of one launch, HIP events on the launch stream."""
    for _ in range(warm):
        fn()
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / reps


def gemm_rooflines(batch, width, reps, dtype="fp32", hidden=0):
    """Live timings of the three Linear contractions at the hidden-layer shape."""
    from bilinear_amd import _native as N
    lib = N.lib()
    gemm = {"bf16x3": lib.blh_gemm_bf16x3}.get(dtype, lib.blh_gemm_f32)
    dev = torch.device("cuda", torch.cuda.current_device())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if dtype == "fp16x2":      # operand maxima once (inside the network the producers supply them)
        ws16 = [torch.empty(lib.blh_gemm_fp16x2_workspace_bytes(), dtype=torch.uint8, device=dev) for _ in range(3)]
        ready = [0, 0, 0]

        def gemm16(slot):
            def call(*a):
                rc = lib.blh_gemm_fp16x2(*a, ws16[slot].data_ptr(), ready[slot])
                ready[slot] = 1
                return rc
            return call
    if dtype == "bf16s":
        return gemm_rooflines_bf16s(batch, width, reps, hidden)
    A = torch.randn(batch, width, device=dev)
    Wt = torch.randn(width, width, device=dev) * 0.03
    bias = torch.randn(width, device=dev)
    Z = torch.empty(batch, width, device=dev)
    stat = torch.empty((batch + 127) // 128, 2, width, device=dev)
    splits = max(1, min((256 * 128 * 128) // (width * width), batch // 128))
    slabs = torch.empty(splits, width, width, device=dev)
    flop = 2.0 * batch * width * width
    out = {}

    def fwd():
        if dtype == "fp16x2":
            N.check(gemm16(0)(st, A.data_ptr(), width, 0, Wt.data_ptr(), width, 0, Z.data_ptr(), width,
                              batch, width, width, 1, bias.data_ptr(), None, 0), "fwd")
        elif dtype != "fp32":
            N.check(gemm(st, A.data_ptr(), width, 0, Wt.data_ptr(), width, 0, Z.data_ptr(), width,
                         batch, width, width, 1, bias.data_ptr(), None, 0), "fwd")
        else:
            N.check(lib.blh_linear_fwd_stats(st, A.data_ptr(), Wt.data_ptr(), bias.data_ptr(),
                                             Z.data_ptr(), stat.data_ptr(), batch, width, width), "fwd")

    def dgrad():
        if dtype == "fp16x2":
            N.check(gemm16(1)(st, A.data_ptr(), width, 0, Wt.data_ptr(), width, 1, Z.data_ptr(), width,
                              batch, width, width, 1, None, None, 0), "dgrad")
            return
        N.check(gemm(st, A.data_ptr(), width, 0, Wt.data_ptr(), width, 1, Z.data_ptr(),
                                 width, batch, width, width, 1, None, None, 0), "dgrad")

    def wgrad():
        if dtype == "fp16x2":
            N.check(gemm16(2)(st, A.data_ptr(), width, 1, Z.data_ptr(), width, 1, slabs.data_ptr(), width,
                              width, width, batch, splits, None, None, 0), "wgrad")
            return
        N.check(gemm(st, A.data_ptr(), width, 1, Z.data_ptr(), width, 1,
                                 slabs.data_ptr(), width, width, width, batch, splits, None, None,
                                 0), "wgrad")

    time_kernel(fwd, 3 * reps)     # clock ramp: the first ~50 ms after idle run 10-15 % slow
    for name, fn in (("linear_fwd", fwd), ("linear_dgrad", dgrad), ("linear_wgrad", wgrad)):
        ms = time_kernel(fn, reps)
        out[name] = {"ms": ms, "tflops": flop / ms / 1e9, "reps": reps}
    return out


def bf16s_kernel_name(batch, width):
    """Which bf16-storage GEMM the library picks for the forward contraction at this shape."""
    try:
        from bilinear_amd import _native as N
        rows = N.lib().blh_gemm_bf16s_tile(batch, width, width, 0, 0, 1, 1)
        cols = N.lib().blh_gemm_bf16s_tile_cols(batch, width, width, 0, 0, 1, 1)
        if rows == 256:
            return "gemm_bf16s_256_kernel"
        if cols == 256:
            return "gemm_bf16s_128x256_kernel"
    except Exception:   # noqa: BLE001  (naming only)
        pass
    return "gemm_bf16s_kernel"


def wgrad_plans(width, batch, stages):
    """(slabs per stage of the batched weight gradient or 0, slabs of the per-stage plan): asked of the
    library (blh_wgrad_plan_bf16s -> api_layout.h), so the timings below follow the step's own plan
    under every tuning knob."""
    from bilinear_amd import _native as N
    batched, per_stage = ctypes.c_int32(0), ctypes.c_int32(1)
    N.check(N.lib().blh_wgrad_plan_bf16s(width, batch, max(1, stages), ctypes.byref(batched),
                                         ctypes.byref(per_stage)), "blh_wgrad_plan_bf16s")
    return int(batched.value), int(per_stage.value)


def gemm_rooflines_bf16s(batch, width, reps, hidden=0):
    """The three contractions of a hidden Linear in bf16 storage (gemm_bf16s_kernel.h); with `hidden` stages
    also the batched weight gradient of all of them in one launch, as the fused step runs it (per stage)."""
    from bilinear_amd import _native as N
    lib = N.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    A = torch.randn(batch, width, device=dev).to(torch.bfloat16)
    Wt = (torch.randn(width, width, device=dev) * 0.03).to(torch.bfloat16)
    bias = torch.randn(width, device=dev)
    Z = torch.empty(batch, width, dtype=torch.bfloat16, device=dev)
    stat = torch.empty((batch + 127) // 128, 2, width, device=dev)
    # batch slabs of the weight gradient as the step plans them (api_layout.h: wgrad_plan_h)
    bs, splits = wgrad_plans(width, batch, hidden)
    slabs = torch.empty(splits, width, width, device=dev)
    flop = 2.0 * batch * width * width

    def fwd():
        N.check(lib.blh_gemm_bf16s(st, A.data_ptr(), width, 0, Wt.data_ptr(), width, 0, Z.data_ptr(), width, 1,
                                   batch, width, width, 1, bias.data_ptr(), None, 0, stat.data_ptr()), "fwd")

    def dgrad():
        N.check(lib.blh_gemm_bf16s(st, A.data_ptr(), width, 0, Wt.data_ptr(), width, 1, Z.data_ptr(), width, 1,
                                   batch, width, width, 1, None, None, 0, None), "dgrad")

    def wgrad():
        N.check(lib.blh_gemm_bf16s(st, A.data_ptr(), width, 1, Z.data_ptr(), width, 1, slabs.data_ptr(), width, 0,
                                   width, width, batch, splits, None, None, 0, None), "wgrad")

    out = {}
    time_kernel(fwd, 3 * reps)
    for name, fn in (("linear_fwd", fwd), ("linear_dgrad", dgrad), ("linear_wgrad", wgrad)):
        ms = time_kernel(fn, reps)
        out[name] = {"ms": ms, "tflops": flop / ms / 1e9, "reps": reps}
    if bs:
        del slabs
        dz = torch.randn(hidden, batch, width, device=dev).to(torch.bfloat16)
        act = torch.randn(hidden, batch, width, device=dev).to(torch.bfloat16)
        outb = torch.empty(hidden, bs, width, width, device=dev)

        def wgrad_batched():
            N.check(lib.blh_gemm_bf16s_batched(st, dz.data_ptr(), width, 1, batch * width, act.data_ptr(), width, 1,
                                               batch * width, outb.data_ptr(), width, bs * width * width,
                                               width, width, batch, hidden, bs), "wgrad batched")
        ms = time_kernel(wgrad_batched, max(20, reps // 8)) / hidden
        out["linear_wgrad_batched"] = {"ms": ms, "tflops": flop / ms / 1e9, "stages": hidden, "slabs": bs,
                                       "note": "per stage, all hidden stages in one launch (the fused step)"}
    return out


def _hbm_record(batch, width, elem_bytes):
    """{kernel name: record} of the newest committed rocprofv3 --pmc record (tools_dev/pmc_hbm.sh ->
    profiles/r06_hbm_traffic.json, else r05) that holds this shape and storage type; {} when none does."""
    for name in ("r06_hbm_traffic.json", "r05_hbm_traffic.json"):
        try:
            with open(os.path.join(REPO, "profiles", name)) as f:
                cfgs = json.load(f)["configs"]
        except (OSError, KeyError, ValueError):
            continue
        for cfg in cfgs.values():
            sh = cfg.get("shape", {})
            if sh.get("B") == batch and sh.get("W") == width and sh.get("s") == elem_bytes:
                return dict(cfg["kernels"], _source="profiles/" + name)
    return {}


def _traffic_sum(rec, prefixes):
    """Sum of traffic_bytes over the kernels named by ``prefixes`` (each a prefix or a (prefix, substring) pair);
    None when one of them is not in the record."""
    total = 0
    for pre in prefixes:
        sub = ""
        if isinstance(pre, tuple):
            pre, sub = pre
        hit = [v["traffic_bytes"] for k, v in rec.items() if k.startswith(pre) and sub in k]
        if not hit:
            return None
        total += hit[0]
    return total or None


def skinny_rooflines_bf16(batch, width, reps):
    """The same for bf16 storage (BASELINE configs[2..4]): the encode stage without Z0 and the one-pass decode as the
    bf16-storage step launches them (blh_skinny_*_bf16), timed live with HIP events.  Algorithmic bytes per pose
    (element size 2): encode forward / backward 32 s + W s + W / 8 (x, A0 or dA0, keep + gate bits), one-pass
    decode 2 W s + 3 * 48 * 4 + 48 * 2 (A, dA; target, pred, dpred in fp32; dpred's bf16 copy).  ``traffic``: PMC bytes
    of the kernels behind the entry point from profiles/r06_hbm_traffic.json when it holds this shape."""
    from bilinear_amd import _native as N
    lib = N.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    IF, OF, W, B = 32, 48, width, batch

    def bf(*shape, scale=1.0):
        return (torch.randn(*shape, device=dev) * scale).to(torch.bfloat16).view(torch.int16)
    x, W0, A, dAin = bf(B, IF), bf(W, IF, scale=0.25), bf(B, W), bf(B, W, scale=1e-3)
    b0 = torch.randn(W, device=dev)
    gam, bet = torch.ones(W, device=dev), torch.zeros(W, device=dev)
    rmean, rvar = torch.zeros(W, device=dev), torch.ones(W, device=dev)
    nbt1 = torch.zeros(1, dtype=torch.int64, device=dev)
    saved = torch.empty(4, W, device=dev)
    scratch = torch.empty(B * W, dtype=torch.int16, device=dev)
    A0 = torch.empty(B, W, dtype=torch.int16, device=dev)
    bits = torch.zeros(((B + 3) // 4) * (W // 8), dtype=torch.int32, device=dev)
    dropd = N.Dropout(None, 1, 0, 0, 0, 0)
    dW0 = torch.empty(W, IF, device=dev)
    db0, dg0, dbe0 = (torch.empty(W, device=dev) for _ in range(3))
    Wd = torch.randn(OF, W, device=dev) * 0.05
    bd = torch.randn(OF, device=dev)
    t = torch.randn(B, OF, device=dev)
    pred, dpred = torch.empty(B, OF, device=dev), torch.empty(B, OF, device=dev)
    dA = torch.empty(B, W, dtype=torch.int16, device=dev)
    wsb = lib.blh_skinny_decode_fused_bf16_workspace_bytes(B, W, OF)
    ws = torch.empty(max(int(wsb), 16), dtype=torch.uint8, device=dev)
    enc_bytes = 2.0 * B * (IF + W) + B * W / 8.0
    ops = [
        ("encode_fused_fwd_bf16 (x -> A0 + keep bits, BatchNorm statistics from the sums of x: 3 launches, no Z0; the "
         "bf16-storage step's encode forward)", enc_bytes,
         lambda: lib.blh_skinny_encode_fused_fwd_bf16(st, x.data_ptr(), W0.data_ptr(), b0.data_ptr(), gam.data_ptr(),
                                                      bet.data_ptr(), rmean.data_ptr(), rvar.data_ptr(), nbt1.data_ptr(),
                                                      0.1, saved.data_ptr(), scratch.data_ptr(), A0.data_ptr(),
                                                      bits.data_ptr(), ctypes.byref(dropd), B, W, IF),
         [("enc_xstats_kernel<unsigned short",), "enc_bn_finalize_kernel<unsigned short>", "enc_fwd_h_kernel"]),
        ("encode_fused_bwd_bf16 (dA0 -> dW0, db0, dgamma, dbeta from dA0, the bits and x: 2 launches; the bf16-storage "
         "step's encode backward)", enc_bytes,
         lambda: lib.blh_skinny_encode_fused_bwd_bf16(st, dAin.data_ptr(), x.data_ptr(), W0.data_ptr(), b0.data_ptr(),
                                                      saved.data_ptr(), bits.data_ptr(), scratch.data_ptr(),
                                                      dW0.data_ptr(), db0.data_ptr(), dg0.data_ptr(), dbe0.data_ptr(),
                                                      B, W, IF),
         ["enc_bwd_h_kernel", "enc_bwd_finish_kernel<unsigned short>"]),
        ("decode_fused_bf16 (Linear %d->48 + MSE + dpred + dA = dP Wd from one read of A; the bf16-storage step's "
         "decode)" % W, B * (2.0 * 2 * W + 3 * OF * 4 + OF * 2),
         lambda: lib.blh_skinny_decode_fused_bf16(st, A.data_ptr(), Wd.data_ptr(), bd.data_ptr(), t.data_ptr(),
                                                  pred.data_ptr(), dpred.data_ptr(), dA.data_ptr(), None, ws.data_ptr(),
                                                  ws.numel(), B, W, OF),
         [("decode_fwd_mse_h_kernel", "true>")]),
    ]
    rec = _hbm_record(B, W, 2)
    out = []
    for name, nbytes, fn, kernels in ops:
        if fn() != 0:          # (a shape this entry point does not serve: the step takes the materialised path there)
            continue
        ms = time_kernel(fn, reps)
        gbs = nbytes / (ms * 1e-3) / 1e9
        pre = [(k[0], "") if isinstance(k, tuple) and len(k) == 1 else k for k in kernels]
        out.append({"op": name, "algorithmic_bytes": nbytes, "avg_us": 1e3 * ms, "achieved": gbs,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "bound": "hbm",
                    "traffic": _traffic_sum(rec, pre), "traffic_source": rec.get("_source")})
    return out


def skinny_rooflines(batch, width, reps):
    """HBM roofline of the skinny projections (BASELINE north_star: "achieved HBM GB/s on the skinny
    32-wide input / output projections"), each exactly as the step launches it (blh_skinny_*), timed
    live with HIP events.  Algorithmic bytes per pose (weights amortised over
    the batch): encode fwd 32 s + W s; decode fwd + MSE W s + 48 s (target) + 2 * 48 s (pred, dpred);
    decode bwd 48 s + W s (reads) + W s (dA); encode wgrad W s + 32 s."""
    from bilinear_amd import _native as N
    lib = N.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    IF, OF, W, B = 32, 48, width, batch
    x = torch.randn(B, IF, device=dev)
    W0 = torch.randn(W, IF, device=dev) * 0.25
    b0 = torch.randn(W, device=dev)
    Z = torch.empty(B, W, device=dev)
    part = torch.empty((B + 63) // 64, 2, W, device=dev)
    A = torch.randn(B, W, device=dev)
    Wd = torch.randn(OF, W, device=dev) * 0.05
    bd = torch.randn(OF, device=dev)
    t = torch.randn(B, OF, device=dev)
    pred, dpred = torch.empty(B, OF, device=dev), torch.empty(B, OF, device=dev)
    loss = torch.zeros((), device=dev)
    dWd, dA = torch.empty(OF, W, device=dev), torch.empty(B, W, device=dev)
    dW0 = torch.empty(W, IF, device=dev)
    wsb = lib.blh_skinny_workspace_bytes(B, W, IF, OF)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    rows = ctypes.c_int32(0)
    # the encode stage as the fp32 step launches it (encode_f32.hip): statistics from the sums of x, no Z0
    from bilinear_amd import _native as N2
    gam, bet = torch.ones(W, device=dev), torch.zeros(W, device=dev)
    rmean, rvar = torch.zeros(W, device=dev), torch.ones(W, device=dev)
    nbt1 = torch.zeros(1, dtype=torch.int64, device=dev)
    saved = torch.empty(4, W, device=dev)
    scratch = torch.empty(B * W, device=dev)
    A0 = torch.empty(B, W, device=dev)
    bits = torch.zeros(((B + 7) // 8) * (W // 4), dtype=torch.int32, device=dev)
    dropd = N2.Dropout(None, 1, 0, 0, 0, 0)
    db0, dg0, dbe0 = (torch.empty(W, device=dev) for _ in range(3))
    ops = [
        ("encode_fused_fwd (x -> A0 + keep bits, BatchNorm statistics from the sums of x: 3 launches, no Z0; fp32 operands: the "
         "exact-fp32 step's encode forward)", 4.0 * B * (IF + W) + B * W / 8.0,
         lambda: lib.blh_skinny_encode_fused_fwd(st, x.data_ptr(), W0.data_ptr(), b0.data_ptr(), gam.data_ptr(),
                                                 bet.data_ptr(), rmean.data_ptr(), rvar.data_ptr(), nbt1.data_ptr(), 0.1,
                                                 saved.data_ptr(), scratch.data_ptr(), A0.data_ptr(), bits.data_ptr(),
                                                 ctypes.byref(dropd), B, W, IF)),
        ("encode_fused_bwd (dA0 -> dW0, db0, dgamma, dbeta from dA0, the bits and x: 2 launches: fp32 operands; the exact-fp32 "
         "step's encode backward)", 4.0 * B * (W + IF) + B * W / 8.0,
         lambda: lib.blh_skinny_encode_fused_bwd(st, A.data_ptr(), x.data_ptr(), W0.data_ptr(), b0.data_ptr(),
                                                 saved.data_ptr(), bits.data_ptr(), scratch.data_ptr(), dW0.data_ptr(),
                                                 db0.data_ptr(), dg0.data_ptr(), dbe0.data_ptr(), B, W, IF)),
        ("encode_fwd (Linear 32->%d + BatchNorm partials)" % W, 4.0 * B * (IF + W),
         lambda: lib.blh_skinny_encode_fwd(st, x.data_ptr(), W0.data_ptr(), b0.data_ptr(), Z.data_ptr(),
                                           part.data_ptr(), ctypes.byref(rows), B, W, IF)),
        ("decode_fwd_mse (Linear %d->48 + MSE loss + dpred)" % W, 4.0 * B * (W + 3 * OF),
         lambda: lib.blh_skinny_decode_fwd_mse(st, A.data_ptr(), Wd.data_ptr(), bd.data_ptr(), t.data_ptr(),
                                               pred.data_ptr(), dpred.data_ptr(), None,
                                               ws.data_ptr(), wsb, B, W, OF)),
        ("decode_fused (Linear %d->48 + MSE + dpred + dA = dP Wd from one read of A; fp32 operands: the exact-fp32 step's decode)" % W,
         4.0 * B * (2 * W + 3 * OF),
         lambda: lib.blh_skinny_decode_fused(st, A.data_ptr(), Wd.data_ptr(), bd.data_ptr(), t.data_ptr(),
                                             pred.data_ptr(), dpred.data_ptr(), dA.data_ptr(), None,
                                             ws.data_ptr(), wsb, B, W, OF)),
        ("decode_bwd (dWd = dP^T A, dA = dP Wd)", 4.0 * B * (OF + 2 * W),
         lambda: lib.blh_skinny_decode_bwd(st, dpred.data_ptr(), A.data_ptr(), Wd.data_ptr(), dWd.data_ptr(),
                                           dA.data_ptr(), ws.data_ptr(), wsb, B, W, OF)),
        ("encode_wgrad (dW0 = dZ^T x)", 4.0 * B * (W + IF),
         lambda: lib.blh_skinny_encode_wgrad(st, Z.data_ptr(), x.data_ptr(), dW0.data_ptr(), ws.data_ptr(),
                                             wsb, B, W, IF)),
    ]
    # HBM-side traffic of the kernels behind each entry point, from the committed rocprofv3 --pmc record of THIS
    # shape (tools_dev/pmc_hbm.sh -> profiles/r06_hbm_traffic.json: 2 * FETCH_SIZE + WRITE_SIZE per launch; slab sums
    # are shared between entry points and not attributed); None where no record covers the shape
    kernels_of = {
        "encode_fused_fwd": ["enc_xstats_kernel", "enc_bn_finalize_kernel", "enc_fwd_kernel"],
        "encode_fused_bwd": ["enc_bwd_kernel", "enc_bwd_finish_kernel"],
        "encode_fwd": ["gemm_f32_ring_kernel<64, 128, 2, 2, 0, 0, 2, 32, 3"],
        "decode_fwd_mse": ["decode_fwd_mse_kernel"],
        "decode_fused": ["decode_fused_kernel"],
        "decode_bwd": ["gemm_f32_ring_kernel<64, 128, 2, 2, 0, 1, 0, 32, 3", "gemm_f32_ring_kernel<64, 128, 2, 2, 1, 1, 0, 32, 3"],
        "encode_wgrad": ["gemm_f32_ring_kernel<128, 32, 4, 1, 1, 1, 0, 32, 3"],
    }
    rec = _hbm_record(B, W, 4)

    def traffic_of(opname):
        return _traffic_sum(rec, kernels_of.get(opname.split(" ")[0], []))

    out = []
    for name, nbytes, fn in ops:
        if fn() != 0:          # (a shape this entry point does not serve: the step takes the other path there)
            continue
        ms = time_kernel(fn, reps)
        gbs = nbytes / (ms * 1e-3) / 1e9
        out.append({"op": name, "algorithmic_bytes": nbytes, "avg_us": 1e3 * ms, "achieved": gbs,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "bound": "hbm",
                    "traffic": traffic_of(name), "traffic_source": rec.get("_source")})
    return out


def _traffic_record(batch, width, kernel_substr, elem_bytes=4):
    """HBM traffic (bytes per launch) of a shipped GEMM from the newest committed rocprofv3 --pmc record that
    has this shape: profiles/r06_hbm_traffic.json (tools_dev/pmc_hbm.sh over this round's binary; its "code" field
    names the commit), else profiles/r04_traffic.json (tools_dev/pmc_r04.sh).  bench.py cannot run the PMC passes
    itself (separate profiler runs); None for shapes that were not profiled."""
    rec = _hbm_record(batch, width, elem_bytes)
    if rec.get("_source", "").endswith("r06_hbm_traffic.json"):
        for kernel, r in rec.items():
            if kernel_substr in kernel:
                return r["traffic_bytes"], rec["_source"]
    try:
        with open(os.path.join(REPO, "profiles", "r04_traffic.json")) as f:
            shape = json.load(f)["shapes"].get("%dx%d" % (batch, width), {})
        for kernel, rec in shape.items():
            if kernel_substr in kernel:
                return rec["traffic_bytes"], "profiles/r04_traffic.json"
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def recorded_traffic(batch, width):
    """fp32 ring GEMM forward (EPI 2 = bias + BatchNorm partials); falls back to the round-2 record."""
    t, src = _traffic_record(batch, width, "gemm_f32_ring_kernel<128, 128, 4, 2, 0, 0, 2")
    if t is not None:
        return t, src
    if (batch, width) != (4096, 1024):
        return None, None
    try:
        with open(os.path.join(REPO, "profiles", "r02_traffic.json")) as f:
            return json.load(f)["kernels"]["linear_fwd"]["traffic_bytes"], "profiles/r02_traffic.json"
    except (OSError, KeyError, ValueError):
        return None, None


def recorded_traffic_bf16s(batch, width):
    """bf16-storage forward GEMM (<ROWK, ROWK, bias + BatchNorm partials, bf16 out> of whichever kernel the
    dispatcher picks at this shape); falls back to the round-3 record."""
    t, src = _traffic_record(batch, width, "<0, 0, 2, true", elem_bytes=2)
    if t is not None:
        return t, src
    try:
        with open(os.path.join(REPO, "profiles", "r03_bf16s_traffic.json")) as f:
            rec = json.load(f)["linear_fwd"].get("%dx%d" % (batch, width))
        if rec and (batch, width) != (8192, 1024):     # (that shape changed kernels in round 4)
            return rec["traffic_bytes"], "profiles/r03_bf16s_traffic.json"
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def recorded_ceiling(batch, width):
    if (batch, width) != (4096, 1024):
        return None
    try:
        with open(os.path.join(REPO, "profiles", "r02_traffic.json")) as f:
            return json.load(f)["kernels"]["linear_fwd"]["shape_ceiling"]
    except (OSError, KeyError, ValueError):
        return None


GEMM_REPS = 500       # launches per HIP-event timing of the dominant kernel (gemm_rooflines)


def roofline_block(args, dom):
    """Roofline of the dominant kernel (the WxW Linear forward GEMM at M = batch)."""
    flop = 2.0 * args.batch * args.width * args.width
    if args.dtype == "fp32":
        traffic, traffic_src = recorded_traffic(args.batch, args.width)
        return {
            "kernel": "gemm_f32_ring_kernel<128,128,4,2,ROWK,ROWK,BIAS_STATS> (Linear %dx%d forward, M=%d)" % (
                args.width, args.width, args.batch),
            "bound": "mfma", "achieved": dom["tflops"], "peak": FP32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": dom["tflops"] / FP32_MFMA_PEAK_TFLOPS,
            "traffic": traffic,
            "traffic_unit": "bytes per launch (rocprofv3 PMC: 2*FETCH_SIZE + WRITE_SIZE, %s)" % traffic_src,
            "avg_launch_ms": dom["ms"], "flop_per_launch": flop,
            "avg_launch_ms_how": "HIP events around %d back-to-back launches on the launch stream; the rocprofv3 "
                                 "--kernel-trace average of the same kernel inside the step is in "
                                 "profiles/r06_final_kernel_stats.md" % dom.get("reps", GEMM_REPS),
            # what a kernel of this launch shape (one 128x128 tile per CU) can reach at all: the
            # same kernel with its loop reduced to the MFMAs, measured (profiles/r02_traffic.json)
            "shape_ceiling": recorded_ceiling(args.batch, args.width),
        }
    if args.dtype == "bf16s":
        traffic, traffic_src = recorded_traffic_bf16s(args.batch, args.width)
        return {
            "kernel": "%s<ROWK,ROWK,BIAS_STATS,bf16 out> (Linear %dx%d forward, M=%d, bf16 storage)" % (
                bf16s_kernel_name(args.batch, args.width), args.width, args.width, args.batch),
            "bound": "mfma", "achieved": dom["tflops"], "peak": BF16_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": dom["tflops"] / BF16_MFMA_PEAK_TFLOPS,
            "traffic": traffic,
            "traffic_unit": "bytes per launch (rocprofv3 PMC: 2*FETCH_SIZE + WRITE_SIZE, %s)" % traffic_src,
            "avg_launch_ms": dom["ms"], "flop_per_launch": flop,
            "algorithmic_bytes_per_launch": 2.0 * (2 * args.batch * args.width + args.width * args.width),
        }
    if args.dtype == "fp16x2":
        peak = BF16_MFMA_PEAK_TFLOPS / 3.0   # f16 MFMA peak = bf16 MFMA peak; three MFMAs per product
        return {
            "kernel": "gemm_f16x2_kernel<ROWK,ROWK,BIAS> (Linear %dx%d forward, M=%d; operands split into "
                      "2 scaled fp16 pieces, 3 f16 MFMAs per product)" % (args.width, args.width, args.batch),
            "bound": "mfma", "achieved": dom["tflops"], "peak": peak,
            "unit": "TFLOP/s (fp32-equivalent = f16 MFMA TFLOP/s / 3)", "frac": dom["tflops"] / peak,
            "traffic": None, "avg_launch_ms": dom["ms"], "flop_per_launch": flop,
            "f16_mfma_tflops_executed": 3.0 * dom["tflops"],
        }
    if args.dtype == "bf16x3":
        # fp32 product from six bf16 MFMAs: priced against the bf16 MFMA peak / 6
        peak = BF16_MFMA_PEAK_TFLOPS / 6.0
        return {
            "kernel": "gemm_split_kernel<128,128,2,2,ROWK,ROWK,BIAS> (Linear %dx%d forward, M=%d; "
                      "operands split into 3 bf16 pieces, 6 bf16 MFMAs per product)" % (
                          args.width, args.width, args.batch),
            "bound": "mfma", "achieved": dom["tflops"], "peak": peak,
            "unit": "TFLOP/s (fp32-equivalent = bf16 MFMA TFLOP/s / 6)", "frac": dom["tflops"] / peak,
            "traffic": None, "avg_launch_ms": dom["ms"], "flop_per_launch": flop,
            "bf16_mfma_tflops_executed": 6.0 * dom["tflops"],
        }
    raise ValueError("no roofline block for dtype %s" % args.dtype)


def alt_mode_block(args, dev, x, t, alt):
    """The same workload with the 1024-wide Linear contractions computed to fp32 accuracy on the
    16-bit matrix cores: "bf16x3" (three-way exact bf16 split, six MFMAs per product) or "fp16x2"
    (two fp16 pieces with per-tensor power-of-two scales, three MFMAs).  Reported beside the
    headline, which stays on the exact-fp32 MFMA path."""
    import bilinear_amd
    out = {}
    preds = {}
    for mode in ("fp32", alt):               # first-step agreement from identical state and dropout
        torch.manual_seed(1)
        n, o, _, _ = bilinear_amd.load(dev, num_blocks=args.blocks, width=args.width, gemm_dtype=mode)
        n.train()
        p, l = n.train_step(o, x, t, max_norm=1.0)
        preds[mode] = (p.detach().double(), float(l.item()))
    d = (preds[alt][0] - preds["fp32"][0]).abs().max().item()
    out["first_step_max_abs_pred_diff_vs_exact_fp32"] = d
    out["first_step_pred_rms"] = preds["fp32"][0].pow(2).mean().sqrt().item()
    out["first_step_loss"] = {"fp32": preds["fp32"][1], alt: preds[alt][1]}
    torch.manual_seed(1)
    net, opt, _, _ = bilinear_amd.load(dev, num_blocks=args.blocks, width=args.width, gemm_dtype=alt)
    net.train()
    for _ in range(args.warmup):
        net.train_step(opt, x, t, max_norm=1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, loss = net.train_step(opt, x, t, max_norm=1.0)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out["ms_per_step"] = 1e3 * el / args.steps
    out["poses_per_s"] = args.batch * args.steps / el
    out["final_loss"] = float(loss.item())
    kern = gemm_rooflines(args.batch, args.width, reps=GEMM_REPS, dtype=alt)
    out["kernels"] = kern
    out["arithmetic"] = {
        "bf16x3": "x = h + m + l exactly (3 bf16 pieces); a*b from 6 bf16 MFMAs, fp32 accumulate; "
                  "dropped terms < 2^-25 |ab|; same parity tests and tolerances as the exact path",
        "fp16x2": "x*2^e = hi + lo (2 fp16 pieces, e from the tensor's largest magnitude); a*b from 3 f16 "
                  "MFMAs, fp32 accumulate, exact unscale; same parity tests and tolerances as the exact path",
    }[alt]
    return out


# BASELINE.json configs[i] -> (blocks, width, per-GPU batch, GEMM arithmetic, n_gpus the config names)
BASELINE_CONFIGS = {
    1: dict(blocks=2, width=1024, batch=4096, dtype="fp32", gpus=1,
            text="2-block width 1024 batch 4096 fp32 on 1xMI355X, fused Linear+BN+ReLU+Dropout fwd/bwd"),
    2: dict(blocks=4, width=1024, batch=16384, dtype="bf16s", gpus=1,
            text="4-block width 1024 batch 16384 bf16 on 1xMI355X, MFMA hidden GEMMs + fused Adam"),
    3: dict(blocks=4, width=1024, batch=8192, dtype="bf16s", gpus=8,
            text="4-block width 1024 batch 65536 bf16, data-parallel 8xMI355X (8192 poses per GPU)"),
    4: dict(blocks=8, width=2048, batch=16384, dtype="bf16s", gpus=8,
            text="8-block width 2048 batch 131072 bf16, 8xMI355X (16384 poses per GPU)"),
}


def workload_label(args, n_devices, rehearsal=None):
    """Names the BASELINE.json config only when the run really is that config.  ``n_devices`` is the number of
    DISTINCT GPUs the ranks hold (comm_block), not WORLD_SIZE; a rehearsal is labelled as one."""
    tail = "" if rehearsal is None else " [REHEARSAL: %s]" % rehearsal
    for idx, c in BASELINE_CONFIGS.items():
        if (args.blocks, args.width, args.batch, args.dtype) == (c["blocks"], c["width"], c["batch"], c["dtype"]):
            per_gpu = "" if (n_devices == c["gpus"] and rehearsal is None) else (
                " [per-GPU shape of the config, run on %d GPU%s]" % (n_devices, "" if n_devices == 1 else "s"))
            return "BASELINE configs[%d]: %s%s%s" % (idx, c["text"], per_gpu, tail)
    return "custom: %d-block width %d, batch %d per GPU, %s%s" % (args.blocks, args.width, args.batch, args.dtype, tail)


def log(msg):
    print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def host_cores():
    """CPU share of this process (the GPU box gives one GPU's share of a bigger host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:   # cgroup v2 quota
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def device_identity(dev):
    """What THIS rank holds, for the ``comm`` block of an N > 1 line: hostname, pid, HIP device index, PCI bus id
    (hipDeviceGetPCIBusId of the runtime torch loaded), UUID and name as torch reports them.  Two ranks hold
    DISTINCT GPUs exactly when their (hostname, pci_bus_id) differ."""
    ident = {"hostname": socket.gethostname(), "pid": os.getpid(), "device_index": int(dev.index or 0),
             "visible": os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES")),
             "pci_bus_id": None, "uuid": None, "name": None}
    try:
        props = torch.cuda.get_device_properties(dev)
        ident["name"] = props.name
        uuid = getattr(props, "uuid", None)
        ident["uuid"] = None if uuid is None else str(uuid)
    except Exception as exc:   # noqa: BLE001  (identity only)
        ident["name"] = "unknown (%s)" % type(exc).__name__
    try:
        hip = ctypes.CDLL("libamdhip64.so.7")     # same soname as the runtime torch loaded: resolves to it
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, ctypes.c_int(ident["device_index"])) == 0:
            ident["pci_bus_id"] = buf.value.decode()
    except (OSError, AttributeError):
        pass
    return ident


def distinct_devices(ranks):
    """Number of distinct GPUs behind a list of device_identity() records: (hostname, PCI bus id), falling back to
    the UUID, then to (hostname, visible-devices string, device index) when the runtime gave neither."""
    keys = set()
    for r in ranks:
        if r.get("pci_bus_id"):
            keys.add((r.get("hostname"), "pci", r["pci_bus_id"]))
        elif r.get("uuid"):
            keys.add((r.get("hostname"), "uuid", r["uuid"]))
        else:
            keys.add((r.get("hostname"), "index", r.get("visible"), r.get("device_index")))
    return len(keys)


def comm_block(ident, rehearse, native=False):
    """COLLECTIVE (every rank calls it): the record that lets an N > 1 line prove what it ran on.  Backend and world
    size as the process group reports them, the RCCL version torch was built against and the one the loaded
    librccl answers with, and every rank's device identity (all-gathered).  ``n_distinct_devices`` is what the
    line's ``n_gpus`` reports; a rehearsal (N ranks time-slicing one GPU over gloo) says so in ``rehearsal``."""
    world = dist.get_world_size()
    ranks = [None] * world
    dist.all_gather_object(ranks, dict(ident, rank=dist.get_rank()))
    backend = str(dist.get_backend())
    out = {"backend": backend, "world_size": world, "ranks": ranks, "n_distinct_devices": distinct_devices(ranks),
           "n_hosts": len({r.get("hostname") for r in ranks}),
           "collectives": "bilinear_amd native RCCL (ncclAllReduce from blh_backward)" if native
                          else "torch.distributed process group",
           "rccl_version_torch_built_with": None, "rccl_version_loaded": None, "rehearsal": None}
    try:
        out["rccl_version_torch_built_with"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:   # noqa: BLE001
        pass
    try:
        rccl = ctypes.CDLL("librccl.so.1")
        v = ctypes.c_int(0)
        if rccl.ncclGetVersion(ctypes.byref(v)) == 0:
            out["rccl_version_loaded"] = int(v.value)
    except (OSError, AttributeError):
        pass
    if rehearse:
        out["rehearsal"] = "%d ranks on %d GPU%s over %s: control flow only, NOT a multi-GPU measurement" % (
            world, out["n_distinct_devices"], "" if out["n_distinct_devices"] == 1 else "s", backend)
    elif backend == "nccl" and out["n_distinct_devices"] != world:
        out["warning"] = "%d ranks share %d device(s): not one rank per GPU" % (world, out["n_distinct_devices"])
    return out


def _free_port():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def self_launch(n):
    """``python bench.py --gpus N`` without a launcher: start the N ranks as children of this
    process (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) BEFORE anything here
    has touched the GPU, let rank 0's JSON line through on stdout, and exit with the children's
    return code.  (Nothing is exec'ed: a process that has initialised HIP must never be replaced.)"""
    rehearse = os.environ.get("BLH_BENCH_REHEARSE") == "1"
    # (device_count() may call hipGetDeviceCount on builds without amdsmi, i.e. initialise the HIP
    #  runtime in THIS process; harmless: the ranks are fresh children and nothing is exec'ed)
    have = n if rehearse else torch.cuda.device_count()     # (a rehearsal does not even count devices here)
    if have < n and not rehearse:
        raise SystemExit("bench.py --gpus %d: this node exposes %d GPU%s (set BLH_BENCH_REHEARSE=1 to "
                         "rehearse the multi-rank control flow on one GPU over gloo)" % (n, have, "" if have == 1 else "s"))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    log("self-launch: %s" % " ".join(cmd))
    # rank 0's JSON line goes to stdout; anything else a library printed there (gloo / RCCL banners)
    # goes to stderr, so that stdout carries exactly the one line of the contract
    # the launcher and its ranks get a session (= process group) of their own, so that whatever happens here — a rank
    # that raised, an interrupt, a launcher that died — every descendant can be signalled by group id and none is
    # left holding the GPU
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1, start_new_session=True)
    rc = 1
    try:
        for line in proc.stdout:
            if line.startswith('{"metric"'):
                sys.stdout.write(line)
                sys.stdout.flush()
            else:
                sys.stderr.write(line)
        rc = proc.wait()
    finally:
        left = reap_process_group(proc.pid)
        if left:
            log("self-launch: %d process(es) of the launch group outlived the launcher and were killed" % left)
            rc = rc or 1
    log("self-launch: children exited with code %d" % rc)
    raise SystemExit(rc)


def reap_process_group(pgid, grace_s=5.0):
    """SIGTERM, then SIGKILL after ``grace_s``, to whatever is still alive in process group ``pgid`` (the session
    self_launch started).  Returns how many processes had to be signalled (0 = the group had exited by itself)."""
    import signal

    def members():
        out = []
        for name in os.listdir("/proc"):
            if not name.isdigit():
                continue
            try:
                with open("/proc/%s/stat" % name) as f:
                    fields = f.read().rsplit(")", 1)[1].split()
                if int(fields[2]) == pgid and fields[0] != "Z":
                    out.append(int(name))
            except (OSError, IndexError, ValueError):
                pass
        return out
    alive = members()
    if not alive:
        return 0
    n = len(alive)
    try:
        os.killpg(pgid, signal.SIGTERM)
    except (ProcessLookupError, PermissionError):
        return n
    deadline = time.time() + grace_s
    while time.time() < deadline and members():
        time.sleep(0.1)
    if members():
        try:
            os.killpg(pgid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
    return n


def pre_ramp(one_step, min_ms, agree=None):
    """Un-timed steps until at least ``min_ms`` of wall time have passed with the GPU busy: the
    first ~50 ms after an idle period run at a lower DVFS state (10-15 % slow), and a driver run of
    20 steps after 5 warm-up steps (27 ms of GPU work) would otherwise be timed inside the ramp.
    The driver's --warmup steps still follow; nothing is removed from the timed region.

    ``agree`` (N > 1): maps this rank's elapsed time to one value every rank sees (the maximum over
    ranks).  Every step holds collectives, so all ranks MUST leave the loop after the same number
    of steps; each rank deciding on its own clock would let one rank run ten steps more than its
    peers and wait in an all-reduce nobody answers."""
    if min_ms <= 0:
        return 0.0, 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while True:
        for _ in range(10):
            one_step()
        n += 10
        torch.cuda.synchronize()
        el = 1e3 * (time.perf_counter() - t0)
        if agree is not None:
            el = agree(el)
        if el >= min_ms:
            return el, n


def config_block(idx, dev, steps, ramp_ms):
    """One BASELINE.json config on this GPU (bf16 storage): whole step timed like the headline
    (pre-ramp, ``steps`` steps bracketed by synchronize), plus the roofline of its dominant kernel."""
    import bilinear_amd
    c = BASELINE_CONFIGS[idx]
    a = argparse.Namespace(blocks=c["blocks"], width=c["width"], batch=c["batch"], dtype=c["dtype"])
    torch.manual_seed(1)
    net, opt, _, _ = bilinear_amd.load(dev, num_blocks=a.blocks, width=a.width, gemm_dtype=a.dtype)
    net.train()
    g = torch.Generator(device=dev).manual_seed(1000)
    x = torch.randn(a.batch, 32, device=dev, generator=g)
    t = torch.randn(a.batch, 48, device=dev, generator=g)

    def one_step():
        return net.train_step(opt, x, t, max_norm=1.0)
    ramp, ramp_steps = pre_ramp(one_step, ramp_ms)
    # three segments of ``steps`` steps, the median one reported (all three are in "segments_ms"): a block is 0.1-0.7 s
    # of GPU time, and one segment of the round-5 record came out 21 % slow between two normal runs on the same box
    # (profiles/r05_INDEX.md: configs[3] shape 1.109 against 0.912 ms)
    seg = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            _, loss = one_step()
        torch.cuda.synchronize()
        seg.append(time.perf_counter() - t0)
    el = sorted(seg)[1]
    final = float(loss.item())
    fwd, bwd = flops_per_pose(a.blocks, a.width)
    poses = a.batch * steps / el
    kern = gemm_rooflines(a.batch, a.width, reps=200, dtype=a.dtype, hidden=2 * a.blocks)
    out = {
        "workload": workload_label(a, 1) + "; x~N(0,1)[B,32], t~N(0,1)[B,48], Kaiming-normal init",
        "value": poses, "unit": "poses/s", "ms_per_step": 1e3 * el / steps, "steps": steps,
        "segments_ms": [1e3 * e / steps for e in seg], "timing": "median of three segments of `steps` steps",
        "pre_ramp_ms": ramp, "dtype": DTYPE_TEXT[a.dtype], "final_loss": final,
        "num_blocks": a.blocks, "width": a.width, "per_gpu_batch": a.batch,
        "step": "zero_grad+forward+MSE+backward+clip_grad_norm(1)+Adam",
        "step_tflops": poses * (fwd + bwd) / 1e12,
        "step_frac_of_bf16_mfma_peak": poses * (fwd + bwd) / 1e12 / BF16_MFMA_PEAK_TFLOPS,
        "roofline": roofline_block(a, kern["linear_fwd"]),
        "roofline_hbm": skinny_rooflines_bf16(a.batch, a.width, reps=200) if a.dtype == "bf16s"
                        else skinny_rooflines(a.batch, a.width, reps=200),
        "kernels": kern,
    }
    del net, opt, x, t
    torch.cuda.empty_cache()
    return out


def batch64_block(dev, steps, ramp_ms):
    """The reference's own batch size (util/config.py:15: 64 poses; BASELINE configs[0]'s shape) on the GPU, fp32,
    2 blocks x 1024, two forms of the same step: one launch per stage (csrc/small_step.hip: the DEFAULT at <= 384 rows)
    and the multi-launch path every larger batch takes; then the reference's five-call loop on the drop-in surface."""
    import bilinear_amd
    torch.manual_seed(1)
    net, opt, _, _ = bilinear_amd.load(dev, num_blocks=2, width=1024, gemm_dtype="fp32")
    net.train()
    net.engine.ensure(dev)
    g = torch.Generator(device=dev).manual_seed(1000)
    x = torch.randn(64, 32, device=dev, generator=g)
    t = torch.randn(64, 48, device=dev, generator=g)

    def one_step():
        return net.train_step(opt, x, t, max_norm=1.0)
    out = {}
    for name, small in (("staged", 1), ("multi_launch", 0)):
        net.engine.set_small_step(small)
        pre_ramp(one_step, ramp_ms)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            _, loss = one_step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out[name] = {"ms_per_step": 1e3 * el / steps, "poses_per_s": 64 * steps / el, "final_loss": float(loss.item())}
    # the reference's five-call step body on the drop-in surface (train_bilinear.py:75-83): one-launch forward and
    # backward kernels, torch's MSELoss, clip and Adam as their own calls — host-bound at this size
    import bilinear_amd as _B
    net.engine.set_small_step(1)
    crit = torch.nn.MSELoss()

    def five_calls():
        opt.zero_grad()
        loss = crit(net(x), t)
        loss.backward()
        _B.clip_grad_norm_(net.parameters(), max_norm=1, module=net)
        opt.step()

    def time_five():
        for _ in range(50):
            five_calls()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            five_calls()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps
    five_ms = time_five()
    # the same loop with torch's backward run on the calling thread (no hand-off to the autograd engine's device
    # thread): a setting of the caller's, not of this package — reported beside the default
    with torch.autograd.set_multithreading_enabled(False):
        five_st_ms = time_five()
    del net, opt
    torch.cuda.empty_cache()
    return {"workload": "2 blocks x 1024, batch 64 (the reference's batch_size), fp32, whole training step",
            "value": out["staged"]["poses_per_s"], "unit": "poses/s",
            "ms_per_step": out["staged"]["ms_per_step"], "steps": steps,
            "launch": "small_step.hip, one launch per stage (the default of the fused step): 2 nh + 3 = 13 launches per step",
            "multi_launch": out["multi_launch"], "final_loss": out["staged"]["final_loss"],
            "five_call_drop_in": {"ms_per_step": five_ms, "poses_per_s": 64e3 / five_ms,
                                  "step": "zero_grad, forward, nn.MSELoss, backward, clip_grad_norm_, Adam.step as "
                                          "separate calls (the reference's loop); host-bound",
                                  "single_threaded_backward_ms_per_step": five_st_ms,
                                  "single_threaded_backward": "the same loop under "
                                                              "torch.autograd.set_multithreading_enabled(False)"}}


DTYPE_TEXT = {"fp32": "f32", "bf16x3": "f32 (operands split into 3 bf16 pieces, bf16 MFMA, fp32 accumulate)",
              "fp16x2": "f32 (operands split into 2 scaled fp16 pieces, f16 MFMA, fp32 accumulate)",
              "bf16s": "bf16 (activations / gradients / weight shadow stored in bf16, bf16 MFMA, fp32 "
                       "accumulate, fp32 master weights + BatchNorm statistics + Adam)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", type=int, choices=sorted(BASELINE_CONFIGS), default=1,
                    help="BASELINE.json configs[i]: sets --blocks/--width/--batch/--dtype (explicit flags "
                         "override); default 1 = the headline (2-block, width 1024, batch 4096, fp32)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch")
    ap.add_argument("--blocks", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=["fp32", "bf16s", "bf16x3", "fp16x2"], default=None,
                    help="GEMM arithmetic: fp32 MFMA (BASELINE configs[1], default), bf16 storage with bf16 "
                         "MFMA and fp32 accumulation (configs[2..4]), or the fp32-accurate split modes")
    ap.add_argument("--sync-bn", action="store_true",
                    help="N>1: BatchNorm statistics over the global batch (exact reference semantics)")
    ap.add_argument("--graph", action="store_true",
                    help="N=1: replay the hipGraph-captured step (bilinear_amd.CapturedTrainStep) "
                         "instead of enqueuing it eagerly; slower at B=4096 (cross-queue graph edges "
                         "cost 10-18 us each, profiles/r02_step_timeline.md) and equal at B=64, so "
                         "eager is the default")
    ap.add_argument("--no-graph", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--graph-two-stream", action="store_true",
                    help="--graph: capture the forked two-stream DAG (default: single-stream order)")
    ap.add_argument("--one-stream", action="store_true",
                    help="A/B: single-stream backward (BLH_OPT_TWO_STREAM = 0)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --batch is the GLOBAL batch, split evenly over the GPUs "
                         "(SURVEY.md 8(d)); default is weak scaling, --batch per GPU")
    ap.add_argument("--no-strong-line", action="store_true",
                    help="with --gpus N > 1: skip the extra strong-scaling measurement")
    ap.add_argument("--strong-global-batch", type=int, default=None,
                    help="with --gpus N > 1: global batch of the strong-scaling sub-measurement (default: the "
                         "config's batch, e.g. 4096 -> 512 rows per rank at N = 8); must divide by 32 * N")
    ap.add_argument("--cpu-steps", type=int, default=12)
    ap.add_argument("--no-alt", action="store_true",
                    help="skip the extra timing of the bf16x3 GEMM mode (reported beside the headline)")
    ap.add_argument("--persistent-shadow", action="store_true", help=argparse.SUPPRESS)     # (the default since r06)
    ap.add_argument("--no-persistent-shadow", action="store_true",
                    help="bf16s A/B: re-cast the fp32 arena to bf16 every step instead of letting Adam write the bf16 "
                         "weight image (Engine.set_persistent_shadow(False); the image is the default since round 6)")
    ap.add_argument("--rehearse-rccl", action="store_true",
                    help="developer: at --gpus 1, run the multi-GPU code path (RCCL group of one rank, "
                         "data-parallel driver with every collective issued) — what the step costs before "
                         "the wire; the line says so in config.parallelism")
    ap.add_argument("--native-rccl", action="store_true",
                    help="data-parallel step with the collectives issued by the library itself "
                         "(DataParallel(collectives='native'): blh_train_step_dp, csrc/comm.hip) instead of "
                         "torch.distributed's process group; opt-in, the line's comm.collectives says which ran")
    ap.add_argument("--native-tail", choices=("producer", "comm"), default="producer",
                    help="with --native-rccl: the stream the last bucket and the optimiser run on")
    ap.add_argument("--no-configs", action="store_true",
                    help="N=1, default config: skip the per-config blocks (configs[2], per-GPU shapes of "
                         "configs[3] and configs[4])")
    ap.add_argument("--config-steps", type=int, default=100, help="timed steps of each per-config block")
    ap.add_argument("--pre-ramp-ms", type=float, default=200.0,
                    help="un-timed steps before --warmup until this much wall time has passed (clock ramp)")
    args = ap.parse_args()
    explicit_shape = any(getattr(args, k) is not None for k in ("blocks", "width", "batch", "dtype"))
    cfg = BASELINE_CONFIGS[args.config]
    for key in ("blocks", "width", "batch", "dtype"):
        if getattr(args, key) is None:
            setattr(args, key, cfg[key])

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)                 # does not return
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.strong:
        if args.batch % (32 * world) != 0:
            raise SystemExit("--strong needs a global batch divisible by 32 * n_gpus")
        args.batch //= world          # per-GPU rows from here on
    # rehearsal on a one-GPU box (developer use): BLH_BENCH_REHEARSE=1 puts every rank on GPU 0
    # and exchanges over gloo, to exercise this file's multi-rank control flow without RCCL
    rehearse = os.environ.get("BLH_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    multi = world > 1 or args.rehearse_rccl
    if args.rehearse_rccl:
        if world != 1:
            raise SystemExit("--rehearse-rccl is a --gpus 1 mode")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import bilinear_amd
    from bilinear_amd.dp import DataParallel

    # Order of set-up under RCCL: the process group first (eagerly: device_id), everything else after it.
    # The order no longer matters for speed (round 3: 2.3-3.0 ms per step with the model first; cause and
    # repair in profiles/r04_dp_setup_order.md: the engine tunes its stream pair); BLH_BENCH_INIT_LAST=1
    # runs the other order.
    init_first = os.environ.get("BLH_BENCH_INIT_LAST") != "1"
    if multi and init_first:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo") if rehearse else dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(1)          # identical init on every rank
    net, opt, step, _ = bilinear_amd.load(dev, num_blocks=args.blocks, width=args.width,
                                          gemm_dtype=args.dtype)
    net.train()
    net.engine.ensure(dev)
    if args.one_stream:
        net.engine.set_two_stream(False)
    args.persistent_shadow = args.dtype == "bf16s" and not args.no_persistent_shadow
    if args.dtype == "bf16s":
        net.engine.set_persistent_shadow(args.persistent_shadow)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    x = torch.randn(args.batch, 32, device=dev, generator=g)
    t = torch.randn(args.batch, 48, device=dev, generator=g)
    if multi:
        net.engine.workspace(args.batch)      # activations / gradient staging for this batch
        opt._ensure_moments(net.engine)       # exp_avg / exp_avg_sq arenas
        strong_pre = None
        strong_global = args.strong_global_batch or cfg["batch"]
        if not args.strong and not args.no_strong_line and strong_global % (32 * world) == 0:
            # (the second model of the strong-scaling sub-measurement, for the same reason)
            torch.manual_seed(1)
            net_s, opt_s, _, _ = bilinear_amd.load(dev, num_blocks=args.blocks, width=args.width,
                                                   gemm_dtype=args.dtype)
            net_s.train()
            net_s.engine.ensure(dev)
            net_s.engine.workspace(strong_global // world)
            opt_s._ensure_moments(net_s.engine)
            strong_pre = (net_s, opt_s)
        torch.cuda.synchronize()
        if not init_first:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo") if rehearse else dist.init_process_group("nccl", device_id=dev)
    if args.native_rccl and rehearse:
        raise SystemExit("--native-rccl needs RCCL (BLH_BENCH_REHEARSE runs over gloo)")
    dp_kw = dict(collectives="native", native_tail=args.native_tail) if args.native_rccl else {}
    dp = DataParallel(net, opt, sync_bn=args.sync_bn, force_collectives=args.rehearse_rccl, **dp_kw) if multi else None
    use_graph = (not multi) and args.graph and not args.no_graph
    captured = None
    if use_graph:
        captured = bilinear_amd.CapturedTrainStep(net, opt, args.batch, max_norm=1.0,
                                                  two_stream=not args.one_stream and args.graph_two_stream,
                                                  persistent_shadow=args.persistent_shadow)
        captured.x.copy_(x)            # the batch lives in the graph's static input buffers
        captured.t.copy_(t)
        x, t = captured.x, captured.t

    dp_captured = None
    if dp is not None and args.graph and not args.no_graph and not args.sync_bn:
        # the data-parallel step as one hipGraph (RCCL calls captured like kernels)
        from bilinear_amd.dp import CapturedDataParallelStep
        dp_captured = CapturedDataParallelStep(dp, args.batch)

    def one_step():
        if dp_captured is not None:
            return dp_captured(x, t)
        if dp is not None:
            return dp.train_step(x, t)
        if captured is not None:
            return captured(x, t)
        return net.train_step(opt, x, t, max_norm=1.0)

    # the data-parallel loops run on the driver's own high-priority stream (dp.stream: hardware queues)
    if dp is not None and dp.stream is not None:
        dp.stream.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.set_stream(dp.stream)
    log("model built, pre-ramp + warm-up")
    if os.environ.get("BLH_BENCH_FAIL_RANK") == str(rank):
        # developer: rehearse a rank that raises (the launcher must end the other ranks and the parent must exit non-zero)
        raise RuntimeError("BLH_BENCH_FAIL_RANK=%d: injected failure on this rank" % rank)
    def slowest_rank(ms):
        v = torch.tensor([ms], device=dev, dtype=torch.float64)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        return float(v.item())
    ramp_ms, ramp_steps = pre_ramp(one_step, args.pre_ramp_ms, slowest_rank if multi else None)
    for _ in range(args.warmup):
        one_step()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pred, loss = one_step()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if multi:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    final_loss = float(loss.item())
    log("timed %d steps: %.3f ms/step" % (args.steps, 1e3 * elapsed / args.steps))

    # Under --gpus N the headline is weak scaling (per-GPU batch fixed).  SURVEY.md 8(d) also asks
    # for the strong-scaling curve of the headline shape (global batch fixed at the config's batch,
    # split N ways): measured right here with the same barrier / max-over-ranks protocol and
    # reported as a sub-object, so one driver run per N yields both curves.
    strong = None
    if multi and strong_pre is not None:
        sb = strong_global // world
        net_s, opt_s = strong_pre
        dp_s = DataParallel(net_s, opt_s, sync_bn=args.sync_bn, force_collectives=args.rehearse_rccl, **dp_kw)
        if sb <= args.batch:
            xs, ts = x[:sb].contiguous(), t[:sb].contiguous()
        else:                                   # (a rehearsal with a tiny --batch)
            xs = torch.randn(sb, 32, device=dev, generator=g)
            ts = torch.randn(sb, 48, device=dev, generator=g)
        n_s = max(20, min(args.steps, 300))
        for _ in range(max(5, min(args.warmup, 50))):
            dp_s.train_step(xs, ts)
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_s):
            dp_s.train_step(xs, ts)
        dist.barrier()
        torch.cuda.synchronize()
        el = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        el = float(el.item())
        strong = {"scaling": "strong", "global_batch": sb * world, "per_gpu_batch": sb, "steps": n_s,
                  "ms_per_step": 1e3 * el / n_s, "value": sb * world * n_s / el, "unit": "poses/s"}
        log("strong scaling (global batch %d): %.3f ms/step" % (sb * world, strong["ms_per_step"]))
        del dp_s, net_s, opt_s

    if dp is not None and dp.stream is not None:
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.default_stream(dev))

    # what the ranks ran on (collective: every rank takes part); N = 1 without a process group has nothing to gather
    comm = comm_block(device_identity(dev), rehearse, native=bool(getattr(args, "native_rccl", False))) \
        if (multi and dist.is_initialized()) else None
    if comm is not None and args.native_rccl and dp is not None:
        # the library's own communicator, as it reports itself: world, RCCL version, collectives issued so far
        comm["native_comm"] = dict(dp.native_comm().info(), tail=args.native_tail)
    n_devices = comm["n_distinct_devices"] if comm is not None else 1
    rehearsal = comm["rehearsal"] if comm is not None else None

    # fwd+bwd only (no optimiser), single rank view, for the record
    def fwd_bwd():
        opt.zero_grad()
        p = net(x)
        l, dpred = net.engine.mse_loss_grad(p.detach(), t)
        p.backward(dpred)
    fb_ms = None
    if rank == 0:
        fb_ms = time_kernel(fwd_bwd, max(5, args.steps // 4), warm=5)

    result = None
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        poses = args.batch * world * args.steps / elapsed
        fwd, bwd = flops_per_pose(args.blocks, args.width)
        log("fwd+bwd only: %.3f ms" % fb_ms)
        # (long enough for the clocks to settle: the first ~50 ms after an idle period run at a
        #  lower DVFS state and read 10-15 % slow)
        kern = gemm_rooflines(args.batch, args.width, reps=GEMM_REPS, dtype=args.dtype, hidden=2 * args.blocks)
        log("kernel timings: %s" % json.dumps(kern))
        dom = kern["linear_fwd"]
        result = {
            "metric": "poses/sec (fwd+bwd, 16-joint, batch 4096) at 1/2/4/8 MI355X",
            # a rehearsal (N ranks time-slicing one GPU over gloo) measures control flow, not throughput: no value
            "value": None if rehearsal else poses,
            "unit": "poses/s",
            # DISTINCT devices the ranks held (comm.ranks[*].pci_bus_id), not WORLD_SIZE
            "n_gpus": n_devices,
            "n_ranks": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": DTYPE_TEXT[args.dtype],
            "pre_ramp_ms": ramp_ms,
            "pre_ramp_steps": ramp_steps,
            "protocol": "untimed pre-ramp (%d steps, %.0f ms: DVFS ramp after idle) + --warmup steps, then --steps "
                        "timed steps between barrier+synchronize pairs; rounds 1-2 had no pre-ramp (their "
                        "20-step driver runs were timed inside the clock ramp): compare across rounds with "
                        "--pre-ramp-ms 0" % (ramp_steps, ramp_ms),
            "data": "synthetic",
            "config": {
                "workload": workload_label(args, n_devices, rehearsal) + "; x~N(0,1)[B,32], t~N(0,1)[B,48], "
                            "Kaiming-normal init",
                "num_blocks": args.blocks, "width": args.width, "per_gpu_batch": args.batch,
                "step": "zero_grad+forward+MSE+backward%s+clip_grad_norm(1)+Adam" % (
                    "+allreduce(grad)" if multi else ""),
                "global_batch": args.batch * world,
                "parallelism": "dp%d%s" % (world, " (RCCL group of one rank, every collective issued)"
                                           if args.rehearse_rccl else (
                                               " (REHEARSAL: %d ranks time-slicing %d GPU over %s)" % (
                                                   world, n_devices, comm["backend"]) if rehearsal else "")),
                "dropout": "philox",
                "batchnorm": ("sync (global batch)" if args.sync_bn else "per-rank statistics") if world > 1 else "single device",
                "launch": ("hipGraph replay (1 launch/step)" if (use_graph or dp_captured is not None)
                           else ("eager, small-batch kernels (one launch per stage: %d stage launches + 3 = %d launches/step)" % (
                                     2 * (1 + 2 * args.blocks), 2 * (1 + 2 * args.blocks) + 3)
                                 if (args.batch <= 384 and args.dtype == "fp32" and not multi)
                                 else "eager (~55 launches/step, weight-gradient GEMMs on a side stream)")) + (
                               "; Adam writes the bf16 weight image (persistent shadow)"
                               if args.persistent_shadow and args.dtype == "bf16s" else ""),
            },
            "final_loss": final_loss,
            "fwd_bwd_only": {"ms_per_step": fb_ms, "poses_per_s": args.batch / (fb_ms / 1e3)},
            "step_tflops": poses * (fwd + bwd) / 1e12,
            ("step_frac_of_bf16_mfma_peak" if args.dtype == "bf16s" else "step_frac_of_fp32_mfma_peak"):
                poses * (fwd + bwd) / 1e12 / ((BF16_MFMA_PEAK_TFLOPS if args.dtype == "bf16s"
                                               else FP32_MFMA_PEAK_TFLOPS) * max(1, n_devices)),
            "roofline": roofline_block(args, dom),
            "roofline_hbm": skinny_rooflines_bf16(args.batch, args.width, reps=300) if args.dtype == "bf16s"
                            else skinny_rooflines(args.batch, args.width, reps=300),
            "kernels": kern,
        }
        if comm is not None:
            result["comm"] = comm
        if rehearsal:
            result["rehearsal"] = rehearsal
            result["control_flow_figure"] = {"poses_per_s_all_ranks_on_shared_gpu": poses,
                                             "note": "ranks time-slice the device(s) listed in comm.ranks: this is "
                                                     "neither a 1-GPU nor an N-GPU throughput"}
        if strong is not None:
            if rehearsal:
                strong = dict(strong, value=None, control_flow_value=strong["value"], rehearsal=rehearsal)
            result["strong_scaling"] = strong
        if world == 1 and args.dtype == "fp32" and not args.no_alt:
            result["fp32_on_16bit_mfma"] = {m: alt_mode_block(args, dev, x, t, m) for m in ("bf16x3", "fp16x2")}
        if world == 1 and args.config == 1 and not explicit_shape and not args.no_configs and not use_graph:
            # the other BASELINE.json configs that fit one GPU, each measured like the headline
            # (the headline's model is released first: configs[4] wants ~3 GB of workspace)
            blocks = {}
            for idx in (2, 3, 4):
                b = config_block(idx, dev, args.config_steps, args.pre_ramp_ms)
                blocks["configs[%d]" % idx] = b
                log("configs[%d]: %.3f ms/step, %.3g poses/s, fwd GEMM %.0f TFLOP/s" % (
                    idx, b["ms_per_step"], b["value"], b["roofline"]["achieved"]))
            result["configs"] = blocks
            result["batch_64"] = batch64_block(dev, 10 * args.config_steps, args.pre_ramp_ms)
            log("batch 64: %.3f ms/step (one launch per stage), %.3f multi-launch" % (
                result["batch_64"]["ms_per_step"], result["batch_64"]["multi_launch"]["ms_per_step"]))
        if world == 1 and not args.no_cpu_baseline:
            from oracle import torch_port as TP
            cores = host_cores()
            log("cpu baseline on %d threads" % cores)
            cpu = TP.time_cpu_steps(args.blocks, args.width, args.batch, steps=args.cpu_steps,
                                    warmup=3, threads=cores)
            # also forward + backward alone, and BASELINE configs[0]'s batch of 64
            # (the reference's own batch size, /root/reference/util/config.py:15)
            cpu_fb = TP.time_cpu_steps(args.blocks, args.width, args.batch, steps=max(10, args.cpu_steps),
                                       warmup=3, threads=cores, fwd_bwd_only=True)
            cpu64 = TP.time_cpu_steps(args.blocks, args.width, 64, steps=50, warmup=5, threads=cores)
            cpu64_fb = TP.time_cpu_steps(args.blocks, args.width, 64, steps=50, warmup=5, threads=cores,
                                         fwd_bwd_only=True)
            result["cpu_baseline"] = {
                "value": cpu["poses_per_s"], "unit": "poses/s", "cores": cpu["threads"],
                "kind": "port",
                "sample": "%d full steps (fwd+MSE+bwd+clip+Adam) of oracle/torch_port.py, batch %d, "
                          "fp32, after 3 warm-up steps" % (cpu["steps"], args.batch),
                "ms_per_step": cpu["ms_per_step"],
                "fwd_bwd": {"value": cpu_fb["poses_per_s"], "unit": "poses/s", "ms_per_step": cpu_fb["ms_per_step"],
                            "sample": "%d x (zero_grad + forward + MSE + backward), batch %d" % (
                                cpu_fb["steps"], args.batch)},
                "batch_64": {"value": cpu64["poses_per_s"], "unit": "poses/s", "ms_per_step": cpu64["ms_per_step"],
                             "fwd_bwd_value": cpu64_fb["poses_per_s"], "fwd_bwd_ms_per_step": cpu64_fb["ms_per_step"],
                             "sample": "50 full steps / 50 forward+backward passes at batch 64 "
                                       "(BASELINE configs[0]), after 5 warm-up steps"},
            }
        print(json.dumps(result), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

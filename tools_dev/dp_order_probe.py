#!/usr/bin/env python3
"""Developer tool: which object's creation order relative to init_process_group("nccl") makes every
kernel of the step slow (profiles/r03_dp_overhead.md item 3)?  One variant per process:

    python3 tools_dev/dp_order_probe.py <variant> [batch]

  group_first      process group, then model / optimizer / workspace           (the prescribed order)
  model_first      model, optimizer state, workspace, one warm step, then the process group
  model_first_realloc   model_first, then every arena and workspace re-allocated after the group exists
                        (module moved to the CPU and back: new tensors, same context and side stream)
  model_first_newctx    model_first, then a NEW library context (side stream, events), same tensors
  model_first_cold      model built first but no kernel launched and no workspace allocated before the group
  tensors_first    only a large torch allocation + one torch kernel before the group, model after it
  no_group         no process group at all (the fused step's reference time)

Prints ms/step of the fused step and of the data-parallel step (world 1, collectives forced) where a
group exists, plus what torch reports about the allocator segments the arenas live in."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bilinear_amd  # noqa: E402
from bilinear_amd.dp import DataParallel  # noqa: E402


def timeit(fn, n=300, warm=150):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def init_group(dev):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29581")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    # (device_id: the communicator is created eagerly, here)
    tt = torch.ones(8, device=dev)
    dist.all_reduce(tt)
    torch.cuda.synchronize()


def build(dev, batch, warm_steps):
    torch.manual_seed(1)
    net, opt, _, _ = bilinear_amd.load(dev)
    net.train()
    x = torch.randn(batch, 32, device=dev)
    t = torch.randn(batch, 48, device=dev)
    for _ in range(warm_steps):
        net.train_step(opt, x, t, max_norm=1.0)
    torch.cuda.synchronize()
    return net, opt, x, t


def main():
    variant = sys.argv[1]
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    out = {}
    if variant == "group_first":
        init_group(dev)
        net, opt, x, t = build(dev, batch, 3)
    elif variant == "no_group":
        net, opt, x, t = build(dev, batch, 3)
    elif variant == "tensors_first":
        big = torch.zeros(64 << 20, device=dev)
        big.add_(1.0)
        torch.cuda.synchronize()
        init_group(dev)
        net, opt, x, t = build(dev, batch, 3)
    elif variant == "model_first_cold":
        torch.manual_seed(1)
        net, opt, _, _ = bilinear_amd.load(dev)
        net.train()
        x = torch.randn(batch, 32, device=dev)
        t = torch.randn(batch, 48, device=dev)
        torch.cuda.synchronize()
        init_group(dev)
    else:
        net, opt, x, t = build(dev, batch, 3)
        out["fused step before the group exists"] = timeit(lambda: net.train_step(opt, x, t, max_norm=1.0), 100, 50)
        init_group(dev)
        if variant == "model_first_realloc":
            sd = {k: v.cpu() for k, v in net.state_dict().items()}
            torch.manual_seed(1)
            net, opt, _, _ = bilinear_amd.load(dev)        # new arenas, new workspace (allocated now)
            net.load_state_dict(sd)
            net.train()
            x, t = x.clone(), t.clone()
        elif variant == "model_first_newctx":
            from bilinear_amd import _native as N
            eng = net.engine
            eng.ctx = N.Context(dev)                        # new context: new events; the side stream is process-wide
        elif variant != "model_first":
            raise SystemExit("unknown variant " + variant)
    out["fused step"] = timeit(lambda: net.train_step(opt, x, t, max_norm=1.0))
    if dist.is_initialized():
        dp = DataParallel(net, opt, force_collectives=True)
        out["dp step (collectives forced)"] = timeit(lambda: dp.train_step(x, t))
        dp.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(dp.stream):
            out["dp step, loop under dp.stream"] = timeit(lambda: dp.train_step(x, t))
        out["fused step, after the dp runs"] = timeit(lambda: net.train_step(opt, x, t, max_norm=1.0), 100, 50)
    for k, v in out.items():
        print("%-22s %-40s %.4f ms/step" % (variant, k, v), flush=True)
    eng = net.engine
    for name in ("params", "grads"):
        ten = getattr(eng, name, None)
        if ten is not None:
            print("%-22s %s at 0x%x, %d bytes" % (variant, name, ten.data_ptr(), ten.numel() * 4))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Developer tool: what does the data-parallel step cost on ONE GPU, against the fused step?
Runs under a world-size-1 RCCL process group:  python3 tools_dev/dp_overhead.py [batch]"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bilinear_amd  # noqa: E402
from bilinear_amd.dp import DataParallel  # noqa: E402


def timeit(fn, n=600, warm=300):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    only = sys.argv[2] if len(sys.argv) > 2 else None      # substring of a variant's name: run only that one
    n, warm = (30, 10) if os.environ.get("DP_SHORT") else (600, 300)   # DP_SHORT=1: for rocprofv3 traces
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    x = torch.randn(batch, 32, device=dev)
    t = torch.randn(batch, 48, device=dev)
    out = {}
    torch.manual_seed(1)
    net, opt, _, _ = bilinear_amd.load(dev)
    net.train()
    if not only:
        out["fused train_step"] = timeit(lambda: net.train_step(opt, x, t, max_norm=1.0))
    for name, kw in (("dp, no collectives (world 1)", dict()),
                     ("dp, collectives forced (RCCL, world 1)", dict(force_collectives=True)),
                     ("dp, forced + bf16 buckets", dict(force_collectives=True, compress="bf16"))):
        if only and only not in name:
            continue
        torch.manual_seed(1)
        net, opt, _, _ = bilinear_amd.load(dev)
        net.train()
        dp = DataParallel(net, opt, **kw)
        out[name] = timeit(lambda: dp.train_step(x, t), n, warm)
        if os.environ.get("DP_UNDER_STREAM"):    # the whole loop on the driver's stream: no hop per step
            dp.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(dp.stream):
                out[name + " [loop under dp.stream]"] = timeit(lambda: dp.train_step(x, t), n, warm)
    if only:
        for k, v in out.items():
            print("%-44s %.4f ms/step" % (k, v))
        dist.destroy_process_group()
        return
    torch.manual_seed(1)
    net, opt, _, _ = bilinear_amd.load(dev)
    net.train()
    out["fused train_step (again, last)"] = timeit(lambda: net.train_step(opt, x, t, max_norm=1.0))
    for k, v in out.items():
        print("%-44s %.4f ms/step" % (k, v))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

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
  model_first_renew        model_first, then blh_side_stream_renew() once the group exists (what DataParallel does)
  model_first_both_before  model_first with the driver's high-priority compute stream ALSO created before the group
  tensors_first_parts      tensors_first, then on the default stream: eval forward, train forward, the fused step
                           single-stream and two-stream; the same again after blh_side_stream_renew()
  tensors_first_streams    tensors_first, then the fused step on the default stream / a fresh normal-priority
                           stream / a fresh high-priority stream, before and after blh_side_stream_renew()

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
    elif variant == "tensors_first_parts":
        big = torch.zeros(64 << 20, device=dev)
        big.add_(1.0)
        torch.cuda.synchronize()
        init_group(dev)
        net, opt, x, t = build(dev, batch, 3)
        from bilinear_amd import _native as N

        def parts(tag):
            net.eval()
            with torch.no_grad():
                out[tag + "eval forward"] = timeit(lambda: net(x), 200, 100)
            net.train()
            out[tag + "fused step, two streams"] = timeit(lambda: net.train_step(opt, x, t, max_norm=1.0), 200, 100)
            net.engine.set_two_stream(False)
            out[tag + "fused step, one stream"] = timeit(lambda: net.train_step(opt, x, t, max_norm=1.0), 200, 100)
            net.engine.set_two_stream(True)
        parts("")
        N.check(N.lib().blh_side_stream_renew(), "blh_side_stream_renew")
        parts("after renew: ")
        for k, v in out.items():
            print("%-22s %-48s %.4f ms/step" % (variant, k, v), flush=True)
        dist.destroy_process_group()
        return
    elif variant == "tensors_first_tune":
        import ctypes
        big = torch.zeros(64 << 20, device=dev)
        big.add_(1.0)
        torch.cuda.synchronize()
        init_group(dev)
        net, opt, x, t = build(dev, batch, 3)
        from bilinear_amd import _native as N

        def probe(tag, stream, cands):
            rep = (ctypes.c_float * 4)()
            N.check(N.lib().blh_tune_streams(ctypes.c_void_p(stream.cuda_stream), cands, rep), "blh_tune_streams")
            print("%-22s %-40s probe: alone %.4f ms, beside current side %.4f (x%.2f), kept %.4f, candidates tried %d" % (
                variant, tag, rep[0], rep[1], rep[1] / rep[0], rep[2], int(rep[3])), flush=True)

        def fused(tag, stream):
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                v = timeit(lambda: net.train_step(opt, x, t, max_norm=1.0), 200, 100)
            torch.cuda.synchronize()
            print("%-22s %-40s fused step %.4f ms" % (variant, tag, v), flush=True)
        d0 = torch.cuda.default_stream(dev)
        sn = torch.cuda.Stream(device=dev)
        sh = torch.cuda.Stream(device=dev, priority=-1)
        for name, st in (("default stream", d0), ("fresh normal-priority stream", sn), ("fresh high-priority stream", sh)):
            probe(name + " (measure only)", st, 0)
            fused(name, st)
        for name, st in (("default stream", d0), ("fresh normal-priority stream", sn), ("fresh high-priority stream", sh)):
            probe(name + " (tune, 3 candidates)", st, 3)
            fused(name + " after its tune", st)
            for n2, s2 in (("default stream", d0), ("fresh normal-priority stream", sn), ("fresh high-priority stream", sh)):
                probe("   then " + n2 + " (measure only)", s2, 0)
        dist.destroy_process_group()
        return
    elif variant == "tensors_first_streams":
        big = torch.zeros(64 << 20, device=dev)
        big.add_(1.0)
        torch.cuda.synchronize()
        init_group(dev)
        net, opt, x, t = build(dev, batch, 3)
        from bilinear_amd import _native as N

        def under(stream):
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                v = timeit(lambda: net.train_step(opt, x, t, max_norm=1.0), 200, 100)
            torch.cuda.synchronize()
            return v
        out["fused, default stream"] = timeit(lambda: net.train_step(opt, x, t, max_norm=1.0), 200, 100)
        out["fused, fresh normal-priority stream"] = under(torch.cuda.Stream(device=dev))
        out["fused, fresh high-priority stream"] = under(torch.cuda.Stream(device=dev, priority=-1))
        N.check(N.lib().blh_side_stream_renew(), "blh_side_stream_renew")
        out["after renew: fused, default stream"] = timeit(lambda: net.train_step(opt, x, t, max_norm=1.0), 200, 100)
        out["after renew: fused, fresh normal-priority stream"] = under(torch.cuda.Stream(device=dev))
        out["after renew: fused, fresh high-priority stream"] = under(torch.cuda.Stream(device=dev, priority=-1))
        for k, v in out.items():
            print("%-22s %-48s %.4f ms/step" % (variant, k, v), flush=True)
        dist.destroy_process_group()
        return
    else:
        if variant == "model_first_both_before":
            from bilinear_amd import dp as _dp
            _dp._COMPUTE_STREAMS[dev] = torch.cuda.Stream(device=dev, priority=-1)
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
        elif variant == "model_first_renew":
            from bilinear_amd import _native as N
            N.check(N.lib().blh_side_stream_renew(), "blh_side_stream_renew")
        elif variant not in ("model_first", "model_first_both_before"):
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

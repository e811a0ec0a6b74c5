"""Worker of tests/test_gpu_dp.py::test_set_up_order_does_not_change_the_step_time (one process per
set-up order): prints ``RESULT <order> <fused ms> <dp ms> <probe alone ms> <probe beside the kept side stream ms>`` for a world-1 RCCL group with every
collective issued.  Orders: group_first | model_first | tensors_first (see the test)."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bilinear_amd  # noqa: E402
from bilinear_amd.dp import DataParallel  # noqa: E402


def timeit(fn, n=200, warm=100):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def main():
    order, port = sys.argv[1], sys.argv[2]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port

    def group():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        tt = torch.ones(8, device=dev)
        dist.all_reduce(tt)
        torch.cuda.synchronize()

    def model():
        torch.manual_seed(1)
        net, opt, _, _ = bilinear_amd.load(dev)
        net.train()
        x = torch.randn(4096, 32, device=dev)
        t = torch.randn(4096, 48, device=dev)
        for _ in range(3):
            net.train_step(opt, x, t, max_norm=1.0)
        torch.cuda.synchronize()
        return net, opt, x, t

    if order == "group_first":
        group()
        net, opt, x, t = model()
    elif order == "model_first":
        net, opt, x, t = model()
        group()
    elif order == "tensors_first":
        big = torch.zeros(16 << 20, device=dev)
        big.add_(1.0)
        torch.cuda.synchronize()
        group()
        net, opt, x, t = model()
    else:
        raise SystemExit("unknown order")
    fused = timeit(lambda: net.train_step(opt, x, t, max_norm=1.0))
    dp = DataParallel(net, opt, force_collectives=True)
    dp.stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(dp.stream):
        dpt = timeit(lambda: dp.train_step(x, t))
    rep = getattr(net.engine, "stream_tune_report", None) or (0.0, 0.0, 0.0, -1.0)
    print("RESULT %s %.4f %.4f %.4f %.4f" % (order, fused, dpt, rep[0], rep[2]), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

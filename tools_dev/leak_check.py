"""developer: create / use / destroy many networks, contexts, captured steps and communicators in one process."""
import gc, os, sys, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bilinear_amd
from bilinear_amd import _native as N
dev = torch.device("cuda", 0)
x = torch.randn(512, 32, device=dev); t = torch.randn(512, 48, device=dev)
def rss(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0
base_alloc = None
for i in range(300):
    dtype = ("fp32", "bf16s")[i % 2]
    net, opt, _, _ = bilinear_amd.load(dev, num_blocks=1, width=256, gemm_dtype=dtype)
    net.train()
    net.train_step(opt, x, t, max_norm=1.0)
    if i % 10 == 0:
        step = bilinear_amd.CapturedTrainStep(net, opt, 512, max_norm=1.0)
        step(x, t)
        del step
    if i % 25 == 0:
        c = N.Comm(dev, N.rccl_unique_id(), 1, 0)
        v = torch.ones(64, device=dev); c.all_reduce(v); torch.cuda.synchronize()
        c.destroy()
    torch.cuda.synchronize()
    del net, opt
    gc.collect()
    if i == 20:
        base_alloc, base_rss = torch.cuda.memory_allocated(), rss()
    if i % 50 == 0:
        print("iteration %3d: device bytes allocated %d, host max RSS %.0f MiB" % (i, torch.cuda.memory_allocated(), rss()), flush=True)
print("device bytes allocated: %d at iteration 20, %d at the end; host max RSS %.0f -> %.0f MiB" % (
    base_alloc, torch.cuda.memory_allocated(), base_rss, rss()))
assert torch.cuda.memory_allocated() <= base_alloc + (1 << 20)
assert rss() <= base_rss + 300, "host memory grew by more than 300 MiB over 280 create / destroy cycles"
print("ok")

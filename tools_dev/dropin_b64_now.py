"""Five-call step at batch 64: ms per step with default autograd threading and with configure_for_small_batches(),
median of 5 segments of 2000 steps each; plus the per-call host split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bilinear_amd
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net, opt, _, _ = bilinear_amd.load(dev)
net.train()
crit = torch.nn.MSELoss()
x = torch.randn(64, 32, device=dev); t = torch.randn(64, 48, device=dev)

def five(n):
    for _ in range(n):
        opt.zero_grad()
        loss = crit(net(x), t)
        loss.backward()
        bilinear_amd.clip_grad_norm_(net.parameters(), max_norm=1, module=net)
        opt.step()

def seg(n=2000):
    torch.cuda.synchronize(); t0 = time.perf_counter(); five(n); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
five(500)
a = sorted(seg() for _ in range(5))
bilinear_amd.configure_for_small_batches()
five(200)
b = sorted(seg() for _ in range(5))
print("five-call step at batch 64: default threading %.3f ms (min %.3f), single-threaded backward %.3f ms (min %.3f)" % (a[2], a[0], b[2], b[0]))

"""The five-call drop-in step at batch 64 with torch's multithreaded backward on (default) and off
(torch.autograd.set_multithreading_enabled(False): backward nodes run on the calling thread, no hand-off to the
autograd engine's device thread)."""
import time
import torch
import bilinear_amd

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net, opt, _, _ = bilinear_amd.load(dev)
net.train()
crit = torch.nn.MSELoss()
x = torch.randn(64, 32, device=dev); t = torch.randn(64, 48, device=dev)


def five():
    opt.zero_grad()
    p = net(x)
    loss = crit(p, t)
    loss.backward()
    bilinear_amd.clip_grad_norm_(net.parameters(), max_norm=1, module=net)
    opt.step()


def timeit(n=2000):
    for _ in range(300):
        five()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        five()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for rep in range(3):
    a = timeit()
    with torch.autograd.set_multithreading_enabled(False):
        b = timeit()
    print("multithreaded backward %.3f ms/step, single-threaded %.3f ms/step" % (a, b), flush=True)

"""cProfile of the five-call drop-in step at batch 64 (host-bound: where do the host microseconds go?)."""
import cProfile
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pstats
import time
import torch
import bilinear_amd

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net, opt, _, _ = bilinear_amd.load(dev)
net.train()
crit = torch.nn.MSELoss()
x = torch.randn(64, 32, device=dev); t = torch.randn(64, 48, device=dev)


def parts():
    out = {}
    def tick(name, t0):
        out[name] = out.get(name, 0.0) + time.perf_counter() - t0
    for _ in range(500):
        t0 = time.perf_counter(); opt.zero_grad(); tick("zero_grad", t0)
        t0 = time.perf_counter(); p = net(x); tick("forward", t0)
        t0 = time.perf_counter(); loss = crit(p, t); tick("mse", t0)
        t0 = time.perf_counter(); loss.backward(); tick("backward", t0)
        t0 = time.perf_counter(); bilinear_amd.clip_grad_norm_(net.parameters(), max_norm=1, module=net); tick("clip", t0)
        t0 = time.perf_counter(); opt.step(); tick("adam", t0)
    torch.cuda.synchronize()
    for k, v in out.items():
        print("%-10s %.1f us" % (k, 1e6 * v / 500))


def five(n):
    for _ in range(n):
        opt.zero_grad()
        p = net(x)
        loss = crit(p, t)
        loss.backward()
        bilinear_amd.clip_grad_norm_(net.parameters(), max_norm=1, module=net)
        opt.step()


import sys
if len(sys.argv) > 1 and sys.argv[1] == "st":
    torch.autograd.set_multithreading_enabled(False)
five(200)
torch.cuda.synchronize()
parts()
pr = cProfile.Profile()
pr.enable()
five(300)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(70)

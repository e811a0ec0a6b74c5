"""Time the reference's five-call step body (zero_grad, forward, MSELoss, backward, clip_grad_norm_, Adam.step) on the
drop-in surface at the reference's batch size, next to the one-call fast path."""
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


def one():
    net.train_step(opt, x, t, max_norm=1.0)


import sys
import bilinear_amd.model.bilinear as MB
if len(sys.argv) > 1:
    MB.EAGER_AUTOGRAD = sys.argv[1]
print("eager autograd bridge:", MB.EAGER_AUTOGRAD)
net.engine.ensure(dev)
if len(sys.argv) > 2:
    net.engine.set_small_step(int(sys.argv[2]))
print("small-step option:", net.engine.ctx.get_option(4))
for name, fn in (("five-call drop-in", five), ("train_step", one)):
    for _ in range(300):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 2000
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("%-20s %.3f ms/step" % (name, 1e3 * el / n))
    # host-only cost: how long the enqueue loop takes when the GPU is not waited for
    t0 = time.perf_counter()
    for _ in range(200):
        fn()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("%-20s host enqueue %.3f ms/step" % (name, 1e3 * host / 200))

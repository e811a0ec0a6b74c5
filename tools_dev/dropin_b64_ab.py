"""Five-call drop-in loop at batch 64: small-step option 1 (persistent halves) against 3 (one launch per stage),
alternating in one process (the loop is host-bound and the host is noisy: compare medians)."""
import statistics, time, torch, bilinear_amd
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net, opt, _, _ = bilinear_amd.load(dev); net.train()
crit = torch.nn.MSELoss()
x = torch.randn(64, 32, device=dev); t = torch.randn(64, 48, device=dev)
def five():
    opt.zero_grad(); loss = crit(net(x), t); loss.backward()
    bilinear_amd.clip_grad_norm_(net.parameters(), max_norm=1, module=net); opt.step()
net.engine.ensure(dev)
res = {1: [], 3: [], 0: []}
for rnd in range(6):
    for mode in (1, 3, 0):
        net.engine.set_small_step(mode)
        for _ in range(100): five()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(800): five()
        torch.cuda.synchronize(); res[mode].append(1e3 * (time.perf_counter() - t0) / 800)
for mode, v in res.items():
    print("option %d: median %.3f  min %.3f  max %.3f ms/step" % (mode, statistics.median(v), min(v), max(v)))

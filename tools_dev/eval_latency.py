import time, torch, bilinear_amd
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net, opt, _, _ = bilinear_amd.load(dev); net.eval()
for B in (1, 64, 256, 384, 512, 1024, 1536, 2048, 4096, 16384):
    x = torch.randn(B, 32, device=dev)
    with torch.no_grad():
        for _ in range(200): net(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2000): net(x)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(200): net(x)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
    print("eval B=%4d: %.1f us per forward (host enqueue %.1f us)" % (B, 1e6 * el / 2000, 1e6 * host / 200))

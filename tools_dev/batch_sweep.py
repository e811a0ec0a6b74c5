"""Fused train step (fp32, 2 x 1024) across batch sizes: ms per step and poses/s."""
import time, torch, bilinear_amd
dev = torch.device("cuda", 0)
for B in (16, 64, 128, 256, 384, 512, 768, 1024, 1280, 1536, 1792, 2048, 2176, 2304, 3072, 4096):
    torch.manual_seed(0)
    net, opt, _, _ = bilinear_amd.load(dev); net.train()
    x = torch.randn(B, 32, device=dev); t = torch.randn(B, 48, device=dev)
    for _ in range(200): net.train_step(opt, x, t)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 1000
    for _ in range(n): net.train_step(opt, x, t)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("B = %5d: %.3f ms/step, %8.0f poses/s" % (B, 1e3 * el / n, B * n / el))
    del net, opt

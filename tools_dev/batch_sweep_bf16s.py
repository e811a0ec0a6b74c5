"""Fused train step (bf16 storage, 4 x 1024) across batch sizes: ms per step and poses/s."""
import time, torch, bilinear_amd
dev = torch.device("cuda", 0)
for B in (1024, 2048, 3072, 4096, 6144, 8192, 10240, 12288, 16384):
    torch.manual_seed(0)
    net, opt, _, _ = bilinear_amd.load(dev, num_blocks=4, gemm_dtype="bf16s"); net.train()
    x = torch.randn(B, 32, device=dev); t = torch.randn(B, 48, device=dev)
    for _ in range(100): net.train_step(opt, x, t)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 400
    for _ in range(n): net.train_step(opt, x, t)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("B = %5d: %.3f ms/step, %9.0f poses/s" % (B, 1e3 * el / n, B * n / el))
    del net, opt

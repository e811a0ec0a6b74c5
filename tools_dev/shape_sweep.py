"""Fused train step across (dtype, blocks, width, batch): looking for anomalies (ms per step, step TFLOP/s)."""
import sys, time, torch, bilinear_amd
dev = torch.device("cuda", 0)
def flops(nb, W, B): return B * (3 * (2 * 32 * W + 2 * nb * 2 * W * W + 2 * W * 48))
for dt, nb, W in (("fp32", 2, 2048), ("fp32", 2, 512), ("bf16s", 8, 2048), ("bf16s", 4, 512)):
    for B in (1024, 2048, 4096, 8192, 16384):
        if dt == "fp32" and B > 8192: continue
        torch.manual_seed(0)
        net, opt, _, _ = bilinear_amd.load(dev, num_blocks=nb, width=W, gemm_dtype=dt); net.train()
        x = torch.randn(B, 32, device=dev); t = torch.randn(B, 48, device=dev)
        for _ in range(30): net.train_step(opt, x, t)
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 100
        for _ in range(n): net.train_step(opt, x, t)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / n
        print("%-5s %d x %4d  B = %5d: %7.3f ms/step  %6.0f TFLOP/s" % (dt, nb, W, B, 1e3 * el, flops(nb, W, B) / el / 1e12))
        del net, opt
        torch.cuda.empty_cache()

"""One training epoch at batch 64 over a device-resident split: data pipeline + train_step, per step."""
import time, torch, bilinear_amd
from bilinear_amd.data import DevicePoseDataset, synthetic_raw
dev = torch.device("cuda", 0)
ds = DevicePoseDataset(synthetic_raw(200000, seed=0), dev)
torch.manual_seed(0)
net, opt, _, _ = bilinear_amd.load(dev); net.train()
for ep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for x, t in ds.epoch(ep, 64, shuffle=True):
        net.train_step(opt, x, t); n += 1
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("epoch %d: %d steps, %.3f ms per step (pipeline + step)" % (ep, n, 1e3 * el / n))

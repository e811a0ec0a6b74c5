"""developer: bf16 storage, Adam's weight image kept across steps, the batch size changing between steps on ONE workspace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bilinear_amd
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
xs = {b: (torch.randn(b, 32, device=dev, generator=g), torch.randn(b, 48, device=dev, generator=g)) for b in (4096, 2048, 1236)}
out = {}
for keep in (True, False):
    torch.manual_seed(0)
    net, opt, _, _ = bilinear_amd.load(dev, num_blocks=2, width=1024, gemm_dtype="bf16s")
    net.train(); net.engine.seed = 5
    net.engine.ensure(dev)
    net.engine.set_persistent_shadow(keep)
    losses = []
    for b in (4096, 4096, 2048, 2048, 4096, 1236, 4096):
        x, t = xs[b]
        losses.append(float(net.train_step(opt, x, t, max_norm=1.0)[1].item()))
    torch.cuda.synchronize()
    out[keep] = (net.engine.params.clone(), losses)
print("image kept:", out[True][1]); print("re-cast   :", out[False][1])
print("parameters bit-identical:", torch.equal(out[True][0], out[False][0]),
      " rel diff %.3e" % ((out[True][0] - out[False][0]).norm() / out[False][0].norm()).item())

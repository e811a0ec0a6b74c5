"""developer: the fused bf16-storage step at ragged batches — loss and decode-bias gradient against what pred / target imply."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bilinear_amd
dev = torch.device("cuda", 0)
shapes = [(1, 256, int(b)) for b in os.environ.get("RAGGED_BATCHES", "129 385 386 388 392 400 1025").split()]
for nb, W, B in shapes:
    torch.manual_seed(0)
    net, opt, _, _ = bilinear_amd.load(dev, num_blocks=nb, width=W, gemm_dtype="bf16s")
    net.train()
    x = torch.randn(B, 32, device=dev); t = torch.randn(B, 48, device=dev)
    pred, loss = net.train_step(opt, x, t, max_norm=1e9)       # (no clipping: the arena keeps the raw gradient)
    torch.cuda.synchronize()
    eng = net.engine
    g = {name: v.clone() for (name, _, _, _), v in zip(eng._named_params(), eng.grad_views())}
    want_loss = ((pred - t) ** 2).mean().item()
    dp = 2.0 * (pred - t) / (B * 48)
    want_db = dp.sum(0)
    got_db = g["decode.bias"]
    print("B %5d: loss %.6f (from pred %.6f)  decode.bias grad rel err %.2e  |got| %.3e |want| %.3e" % (
        B, loss.item(), want_loss, ((got_db - want_db).norm() / want_db.norm()).item(), got_db.norm().item(), want_db.norm().item()), flush=True)

"""developer: forward_train_loss (loss finalised right behind the forward) then backward(dpred = None) at a ragged batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bilinear_amd
dev = torch.device("cuda", 0)
for B in (129, 385, 392, 1025):
    torch.manual_seed(0)
    net, opt, _, _ = bilinear_amd.load(dev, num_blocks=1, width=256, gemm_dtype="bf16s")
    net.train()
    eng = net.engine
    x = torch.randn(B, 32, device=dev); t = torch.randn(B, 48, device=dev)
    pred, loss = eng.forward_train_loss(x, t)
    torch.cuda.synchronize()
    want = ((pred - t) ** 2).mean().item()
    eng.backward(x, None)
    torch.cuda.synchronize()
    g = {name: v.clone() for (name, _, _, _), v in zip(eng._named_params(), eng.grad_views())}
    dp = 2.0 * (pred - t) / (B * 48)
    want_db = dp.sum(0)
    print("B %5d: forward_train_loss %.6f (from pred %.6f); backward(None): decode.bias rel err %.2e" % (
        B, loss.item(), want, ((g["decode.bias"] - want_db).norm() / want_db.norm()).item()), flush=True)
    # the drop-in path
    opt.zero_grad()
    p2 = net(x)
    l2 = torch.nn.functional.mse_loss(p2, t)
    l2.backward()
    torch.cuda.synchronize()
    db = dict(net.named_parameters())["decode.bias"].grad
    dp2 = 2.0 * (p2.detach() - t) / (B * 48)
    print("          drop-in: loss %.6f, decode.bias rel err %.2e" % (l2.item(), ((db - dp2.sum(0)).norm() / dp2.sum(0).norm()).item()), flush=True)

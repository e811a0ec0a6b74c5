"""Gradients of a batch made of c copies of a small batch vs the small batch itself (bf16s): which batch sizes agree?"""
import sys
import numpy as np, torch
sys.path.insert(0, "tests")
import test_gpu_timed_path as T
dev = torch.device("cuda", 0)
nb, width, small = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "bf16s"
st0 = T._state(nb, width, 300 + nb)
g = torch.Generator(device=dev).manual_seed(77)
x = torch.randn(small, 32, device=dev, generator=g); t = torch.randn(small, 48, device=dev, generator=g)
nh = 1 + 2 * nb
masks = [(torch.rand(small, width, device=dev, generator=g) < 0.5).to(torch.uint8) for _ in range(nh)]
ref = None
for rep in (1, 2, 4, 8):
    net, opt = T._build(st0, dev, nb, width, mode)
    net.engine.set_dropout_masks([m.repeat(rep, 1) for m in masks])
    opt.zero_grad()
    pred = net(x.repeat(rep, 1)); loss = torch.nn.functional.mse_loss(pred, t.repeat(rep, 1)); loss.backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    if ref is None:
        ref = grads
    else:
        errs = {k: float((grads[k] - ref[k]).double().norm() / ref[k].double().norm()) for k in grads if not T.is_prebn_bias(k)}
        ks = ["decode.weight", "bilinear.%d.1.1.weight" % (nb - 1), "bilinear.%d.1.0.weight" % (nb - 1), "bilinear.%d.0.1.weight" % (nb - 1), "bilinear.0.0.0.weight", "encode.0.weight"]
        print("B = %6d (%d copies): " % (small * rep, rep) + ", ".join("%s %.1e" % (k.replace("bilinear.", "b"), errs[k]) for k in ks))
    del net, opt

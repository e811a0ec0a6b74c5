"""Diagnosis: encode weight gradient, encode stage without Z0 against the materialised path, torch-initialised net."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bilinear_amd  # noqa: E402


def run(zero_bias, explicit):
    dev = torch.device("cuda:0")
    batch, nb, width = 4096, 2, 1024
    x = torch.randn(batch, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    t = torch.randn(batch, 48, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    g = torch.Generator(device=dev).manual_seed(9)
    masks = [(torch.rand(batch, width, device=dev, generator=g) < 0.5).to(torch.uint8) for _ in range(1 + 2 * nb)]
    out = {}
    for fused in (True, False):
        if not fused:
            os.environ["BLH_NO_ENCODE_FUSE"] = "1"
        else:
            os.environ.pop("BLH_NO_ENCODE_FUSE", None)
        os.environ["BLH_NO_DECODE_FUSE"] = "1"
        torch.manual_seed(0)
        net, opt, _, _ = bilinear_amd.load(dev, num_blocks=nb, width=width, gemm_dtype="fp32")
        net.train()
        net.engine.ensure(dev)
        net.engine.seed = 11
        if zero_bias:
            with torch.no_grad():
                net.encode[0].bias.zero_()
        if explicit:
            net.engine.set_dropout_masks(masks)
        opt.zero_grad()
        pred = net(x)
        torch.nn.functional.mse_loss(pred, t).backward()
        torch.cuda.synchronize()
        out[fused] = {k: p.grad.detach().double().clone() for k, p in net.named_parameters()}
        out[fused]["pred"] = pred.detach().double().clone()
        out[fused]["b0"] = net.encode[0].bias.detach().double().clone()
    print("zero_bias", zero_bias, "explicit masks", explicit, "|b0| max", float(out[True]["b0"].abs().max()))
    for k in ("pred", "encode.0.weight", "encode.1.weight", "encode.1.bias", "bilinear.0.0.0.weight"):
        a, b = out[True][k], out[False][k]
        print("   %-24s rel %.3e" % (k, float((a - b).norm() / b.norm())))
    a, b = out[True]["encode.0.weight"], out[False]["encode.0.weight"]
    d = a - b
    print("   per-feature rel:", [round(float(d[:, f].norm() / b[:, f].norm()), 5) for f in range(0, 32, 4)])
    print("   corr of diff rows with b0:", float(torch.corrcoef(torch.stack([d.norm(dim=1), out[True]["b0"].abs()]))[0, 1]))


for zb in (False, True):
    for ex in (False, True):
        run(zb, ex)

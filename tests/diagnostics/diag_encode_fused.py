"""Diagnosis (uses the oracle, so it lives under tests/): raw gradients of the drop-in step at B = 4096 with the
encode stage without Z0 + one-pass decode, and with both switched off, each against the fp64 oracle on the SAME
(unedited Philox) masks: relative L2 per tensor.  usage: python tests/diagnostics/diag_encode_fused.py [batch]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import test_gpu_timed_path as T  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    dev = torch.device("cuda:0")
    nb, width = 2, 1024
    entry = T._entry_with_masks(nb, width, batch, dev, safe=False)
    r = T._run_oracle(entry, entry["philox"])
    xt, tt = torch.from_numpy(entry["x"]).to(dev), torch.from_numpy(entry["t"]).to(dev)
    res = {}
    for fused in (True, False):
        if not fused:
            os.environ["BLH_NO_ENCODE_FUSE"] = "1"
            os.environ["BLH_NO_DECODE_FUSE"] = "1"
        net, opt = T._build(entry["st0"], dev, nb, width, "fp32")
        net.engine.set_dropout_masks(entry["philox"])
        opt.zero_grad()
        p = net(xt)
        torch.nn.functional.mse_loss(p, tt).backward()
        torch.cuda.synchronize()
        res[fused] = {k: q.grad.detach().cpu().numpy().astype(np.float64) for k, q in net.named_parameters()}
        res[fused]["pred"] = p.detach().cpu().numpy().astype(np.float64)
    ref = dict(r["grads_raw"])
    ref["pred"] = r["pred"]
    for k in res[True]:
        g = np.asarray(ref[k], np.float64)
        n = np.linalg.norm(g) + 1e-300
        print("%-28s |ref| %.3e  fused-vs-oracle %.2e  materialised-vs-oracle %.2e  fused-vs-materialised %.2e" % (
            k, n, np.linalg.norm(res[True][k] - g) / n, np.linalg.norm(res[False][k] - g) / n,
            np.linalg.norm(res[True][k] - res[False][k]) / n))


if __name__ == "__main__":
    main()

"""Developer diagnostic: per-parameter relative L2 error of the raw gradients against the fp64
oracle, for several batch sizes / schedules / dropout sources."""
import sys, os, ctypes
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))        # tests/
sys.path.insert(0, os.path.dirname(_TESTS))                                   # repository root
sys.path.insert(0, _TESTS)
import numpy as np, torch
from oracle import numpy_oracle as O
import test_gpu_timed_path as T

dev = torch.device("cuda:0")
nb, width = 2, 1024
for batch in [int(a) for a in sys.argv[1:]] or [2048, 4096]:
    st = T._state(nb, width, 100 + nb)
    x, t = O.synthetic_batch(5, batch)
    xt, tt = torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev)
    ref = None
    for variant in ("philox-two", "philox-one", "explicit-two"):
        net, opt = T._build(st, dev, nb, width, "fp32")
        masks = T._philox_masks(net, 0, batch)
        if ref is None:
            s2 = {k: v.copy() for k, v in st.items()}
            rp, cache = O.forward(s2, x, masks, training=True, dtype=np.float64)
            rl, dp = O.mse_loss(rp, t.astype(np.float64))
            ref = O.backward(s2, cache, dp, dtype=np.float64)
            # fp32 numpy oracle for scale: what plain fp32 arithmetic gives
            s3 = {k: v.copy() for k, v in st.items()}
            rp32, cache32 = O.forward(s3, x, masks, training=True, dtype=np.float32)
            _, dp32 = O.mse_loss(rp32, t)
            ref32 = O.backward(s3, cache32, dp32, dtype=np.float32)
        if variant == "philox-one":
            net.engine.set_two_stream(False)
        if variant == "explicit-two":
            net.engine.set_dropout_masks(masks)
        opt.zero_grad()
        pred = net(xt)
        loss = torch.nn.functional.mse_loss(pred, tt)
        loss.backward()
        torch.cuda.synchronize()
        print("B=%d %s: pred rel %.2e" % (batch, variant, np.linalg.norm(pred.detach().cpu().numpy() - rp) / np.linalg.norm(rp)))
        for k, p in net.named_parameters():
            g = p.grad.cpu().numpy().astype(np.float64)
            den = np.linalg.norm(ref[k])
            e = np.linalg.norm(g - ref[k]) / den
            e32 = np.linalg.norm(ref32[k].astype(np.float64) - ref[k]) / den
            print("   %-28s gpu %.2e   numpy-fp32 %.2e   |ref| %.2e" % (k, e, e32, den))

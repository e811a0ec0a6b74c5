#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Runs only in the build container, where ``/root/reference`` is mounted
(read-only); the reference never travels, the ``.npz`` files do.  It imports
``/root/reference/model/bilinear.py`` unmodified, loads a deterministic initial
state (``oracle.numpy_oracle.init_state`` — NumPy legacy RandomState, so the
GPU box can regenerate the identical arrays without torch's RNG), and replays
the step body of ``/root/reference/train_bilinear.py:66-83`` (lr-decay check,
zero_grad, forward, MSELoss, backward, clip_grad_norm_(1), Adam.step) for a few
steps, recording inputs, recovered dropout masks and every observable.

Large tensors (the 1024x1024 weights) are recorded as float64 L2 norms plus a
fixed random sample of elements; small ones in full.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(1, "/root/reference")
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

import torch  # noqa: E402
from torch import nn  # noqa: E402

import model.bilinear as ref  # noqa: E402  (the reference)
from oracle import numpy_oracle as O  # noqa: E402

N_SAMPLE = 256
FULL_LIMIT = 2048    # tensors up to this many elements are stored in full


def sample_index(key, numel):
    rng = np.random.RandomState(abs(hash_str(key)) % (2 ** 31))
    return rng.randint(0, numel, size=N_SAMPLE).astype(np.int64)


def hash_str(s):
    h = 2166136261
    for ch in s.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def record_tensor(out, prefix, key, arr, force_full=False):
    arr = np.array(arr, copy=True, order="C")   # torch .numpy() aliases live state
    flat = arr.reshape(-1)
    out["%s/%s/norm" % (prefix, key)] = np.float64(np.sqrt((flat.astype(np.float64) ** 2).sum()))
    out["%s/%s/sum" % (prefix, key)] = np.float64(flat.astype(np.float64).sum())
    if flat.size <= FULL_LIMIT or force_full:
        out["%s/%s/full" % (prefix, key)] = arr
    else:
        out["%s/%s/sample" % (prefix, key)] = flat[sample_index(key, flat.size)]


def run(batch, steps, seed_init, seed_data, fname):
    torch.manual_seed(1000 + batch)
    torch.set_num_threads(8)
    net = ref.BilinearUnit()
    st0 = O.init_state(seed_init, 2, 1024)
    sd = net.state_dict()
    assert list(sd.keys()) == list(st0.keys())
    net.load_state_dict({k: torch.from_numpy(st0[k].copy()).reshape(sd[k].shape) for k in sd})
    opt = torch.optim.Adam(net.parameters(), lr=1.0e-3)     # model/bilinear.py:60
    criterion = nn.MSELoss()                                  # train_bilinear.py:49
    net.train()                                               # train_bilinear.py:54

    drops = [m for m in net.modules() if isinstance(m, nn.Dropout)]
    assert len(drops) == 5
    captured = {}

    def make_hook(i):
        def hook(mod, inp, outp):
            if not mod.training:
                return
            a, b = inp[0].detach(), outp.detach()
            keep = (b != 0)
            # where the input is exactly 0 the mask is unobservable and irrelevant
            keep = torch.where(a > 0, keep, torch.ones_like(keep))
            ok = torch.where(a > 0, (b == 0) | (b == 2 * a), torch.ones_like(keep))
            assert bool(ok.all()), "dropout output not in {0, 2x}"
            captured[i] = keep.numpy().astype(np.uint8)
        return hook

    for i, d in enumerate(drops):
        d.register_forward_hook(make_hook(i))

    out = {}
    out["meta/batch"] = np.int64(batch)
    out["meta/steps"] = np.int64(steps)
    out["meta/seed_init"] = np.int64(seed_init)
    out["meta/seed_data"] = np.int64(seed_data)
    out["meta/torch_version"] = np.array(torch.__version__)
    for k, v in st0.items():
        f = v.reshape(-1).astype(np.float64)
        out["init/%s/sum" % k] = np.float64(f.sum())
        out["init/%s/head" % k] = v.reshape(-1)[:4].copy()

    names = list(dict(net.named_parameters()).keys())
    step = 1                                                  # model/bilinear.py:61
    for s in range(steps):
        x, t = O.synthetic_batch(seed_data + s, batch)
        if O.lr_decay_condition(step):                        # train_bilinear.py:66-70
            lr = O.lr_decay_function(step)
            for pg in opt.param_groups:
                pg["lr"] = lr
        out["step%d/lr" % s] = np.float64(opt.param_groups[0]["lr"])
        xt, tt = torch.from_numpy(x), torch.from_numpy(t)
        opt.zero_grad()                                       # :75
        pred = net(xt)                                        # :76
        loss = criterion(pred, tt)                            # :78
        loss.backward()                                       # :79
        raw = {n: p.grad.detach().numpy().copy() for n, p in net.named_parameters()}
        total_norm = nn.utils.clip_grad_norm_(net.parameters(), max_norm=1)   # :81
        clipped = {n: p.grad.detach().numpy().copy() for n, p in net.named_parameters()}
        opt.step()                                            # :83
        step += 1

        out["step%d/x" % s] = x
        out["step%d/t" % s] = t
        for i in range(5):
            out["step%d/mask%d" % (s, i)] = np.packbits(captured[i], axis=1)
        out["step%d/pred" % s] = pred.detach().numpy().copy()
        out["step%d/loss" % s] = np.float64(loss.item())
        out["step%d/total_norm" % s] = np.float64(float(total_norm))
        for n in names:
            record_tensor(out, "step%d/grad_raw" % s, n, raw[n],
                          force_full=(s == 0 and n in ("encode.0.weight", "decode.weight")))
            record_tensor(out, "step%d/grad_clipped" % s, n, clipped[n])
        sd = net.state_dict()
        for k, v in sd.items():
            record_tensor(out, "step%d/state" % s, k, v.detach().numpy())
        for n, p in net.named_parameters():
            stp = opt.state[p]
            record_tensor(out, "step%d/exp_avg" % s, n, stp["exp_avg"].numpy())
            record_tensor(out, "step%d/exp_avg_sq" % s, n, stp["exp_avg_sq"].numpy())

    # eval-mode forward (valid_bilinear.py:31,37,52)
    net.eval()
    xe, _ = O.synthetic_batch(seed_data + 100, batch)
    with torch.set_grad_enabled(False):
        pe = net(torch.from_numpy(xe))
    out["eval/x"] = xe
    out["eval/pred"] = pe.numpy().copy()

    path = os.path.join(HERE, fname)
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KiB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    run(batch=8, steps=3, seed_init=11, seed_data=21, fname="ref_b8_s3.npz")
    run(batch=64, steps=2, seed_init=12, seed_data=22, fname="ref_b64_s2.npz")

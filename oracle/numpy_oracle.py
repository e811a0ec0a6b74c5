"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

CPU restatement (NumPy) of the one hot path this repository accelerates: the
forward / backward / optimiser step of the 2D->3D pose-lifting MLP of
nulledge/bilinear.  It exists only to *check* the HIP path (tests/,
``__graft_entry__.smoke()``, and the ``cpu_baseline`` leg of ``bench.py`` may
import it; nothing under ``bilinear_amd/`` may).

Parity status: PINNED.  Every function below is checked against golden vectors
captured from the reference itself (``/root/reference/model/bilinear.py`` +
the step body of ``train_bilinear.py``), see ``tests/golden/make_golden.py``
and ``tests/test_oracle_golden.py``.

The arithmetic of the reference lives in PyTorch (un-vendored, un-pinned
dependency; the golden vectors were produced with torch 2.10.0 CPU).  Each
function cites the reference call-site it restates (paths relative to
``/root/reference``) and, where the semantics are PyTorch's, the published
formula it follows.

The restatement is generalised to ``(num_blocks, width)``; the reference only
expresses ``(2, 1024)`` (``model/bilinear.py:22-29``), which is where it is
pinned.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

NUM_JOINT = 17 - 1            # model/bilinear.py:20
IN_FEATURES = 2 * NUM_JOINT   # model/bilinear.py:22
OUT_FEATURES = 3 * NUM_JOINT  # model/bilinear.py:29
BN_EPS = 1e-5                 # nn.BatchNorm1d default, model/bilinear.py:10
BN_MOMENTUM = 0.1             # nn.BatchNorm1d default
DROPOUT_P = 0.5               # model/bilinear.py:12
ADAM_BETAS = (0.9, 0.999)     # torch.optim.Adam default, model/bilinear.py:60
ADAM_EPS = 1e-8
CLIP_MAX_NORM = 1.0           # train_bilinear.py:81


# --------------------------------------------------------------------------
# structure
# --------------------------------------------------------------------------
def heavy_names(num_blocks):
    """Prefixes of every heavy_linear (Linear->BN->ReLU->Dropout) in forward
    order: ``encode`` then ``bilinear.{b}.{l}`` (model/bilinear.py:22-27)."""
    names = ["encode"]
    for b in range(num_blocks):
        for l in range(2):
            names.append("bilinear.%d.%d" % (b, l))
    return names


def state_spec(num_blocks=2, width=1024):
    """Ordered (key, shape, kind) of the reference ``state_dict`` —
    37 entries at num_blocks=2 (22 parameters + 15 BN buffers)."""
    spec = []
    for i, h in enumerate(heavy_names(num_blocks)):
        fan_in = IN_FEATURES if i == 0 else width
        spec += [
            (h + ".0.weight", (width, fan_in), "param"),
            (h + ".0.bias", (width,), "param"),
            (h + ".1.weight", (width,), "param"),
            (h + ".1.bias", (width,), "param"),
            (h + ".1.running_mean", (width,), "buffer"),
            (h + ".1.running_var", (width,), "buffer"),
            (h + ".1.num_batches_tracked", (), "buffer"),
        ]
    spec += [
        ("decode.weight", (OUT_FEATURES, width), "param"),
        ("decode.bias", (OUT_FEATURES,), "param"),
    ]
    return spec


def param_keys(num_blocks=2):
    """Keys in ``module.parameters()`` order (what Adam / clip iterate)."""
    return [k for k, _, kind in state_spec(num_blocks, 4) if kind == "param"]


def init_state(seed, num_blocks=2, width=1024, dtype=np.float32):
    """Deterministic initial state with the *distributions* of
    model/bilinear.py:86-90: Linear.weight ~ kaiming_normal (fan_in, gain
    sqrt 2 => std = sqrt(2/fan_in)); Linear.bias keeps nn.Linear's default
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)); BN gamma=1, beta=0, running stats
    (0, 1), num_batches_tracked=0.  The bit stream is NumPy's legacy
    ``RandomState`` (stable across NumPy versions), not torch's."""
    rng = np.random.RandomState(seed)
    st = OrderedDict()
    for key, shape, _ in state_spec(num_blocks, width):
        leaf = key.split(".")[-1]
        is_linear = key.startswith("decode") or key.split(".")[-2] == "0"
        if is_linear and leaf == "weight":
            fan_in = shape[1]
            st[key] = (rng.standard_normal(shape) * math.sqrt(2.0 / fan_in)).astype(dtype)
        elif is_linear and leaf == "bias":
            fan_in = IN_FEATURES if key.startswith("encode") else width
            bound = 1.0 / math.sqrt(fan_in)
            st[key] = rng.uniform(-bound, bound, size=shape).astype(dtype)
        elif leaf == "weight":          # BN gamma
            st[key] = np.ones(shape, dtype)
        elif leaf == "bias":            # BN beta
            st[key] = np.zeros(shape, dtype)
        elif leaf == "running_mean":
            st[key] = np.zeros(shape, dtype)
        elif leaf == "running_var":
            st[key] = np.ones(shape, dtype)
        else:
            st[key] = np.zeros((), np.int64)
    return st


def synthetic_batch(seed, batch, dtype=np.float32):
    """x ~ N(0,1) [B,32], t ~ N(0,1) [B,48]: the contract of the real loader is
    per-feature z-scored vectors (H36M/data.py:108-110)."""
    rng = np.random.RandomState(seed)
    x = rng.standard_normal((batch, IN_FEATURES)).astype(dtype)
    t = rng.standard_normal((batch, OUT_FEATURES)).astype(dtype)
    return x, t


def random_masks(seed, batch, num_blocks=2, width=1024):
    """Bernoulli(0.5) keep-masks, one [B,width] uint8 array per heavy_linear."""
    rng = np.random.RandomState(seed)
    return [(rng.random_sample((batch, width)) >= DROPOUT_P).astype(np.uint8)
            for _ in heavy_names(num_blocks)]


# --------------------------------------------------------------------------
# GEMM operand rounding (gemm_dtype = "bf16" of the build: tensors stay fp32, the two
# operands of every Linear contraction are rounded to bfloat16, products and sums are exact /
# fp32).  Not part of the reference; restated here so that the bf16 mode of the HIP path can be
# checked against the SAME arithmetic instead of only "close to fp32".
# --------------------------------------------------------------------------
_GEMM_ROUND = None


def set_gemm_rounding(mode):
    """mode None (fp32/fp64 exact operands), "bf16" (the two operands of every contraction are
    rounded to bfloat16, round-to-nearest-even; every tensor is otherwise kept exact — the build's
    gemm_dtype "bf16") or "bf16s" (bf16 STORAGE, the build's gemm_dtype "bf16s": besides the
    operands, every tensor the build keeps in bf16 is rounded where the build stores it — the
    network input, the pre-BN output z (AFTER the batch statistics were taken from the un-rounded
    values), the activation a (after ReLU, dropout and the residual add), the gradient of the
    prediction as the decode contractions read it, and the gradients dA / dZ)."""
    global _GEMM_ROUND
    assert mode in (None, "bf16", "bf16s")
    _GEMM_ROUND = mode


def _round_bf16_reference(a):
    """The defining form (64-bit intermediate): round to nearest even on the upper 16 bits of the fp32 pattern."""
    a32 = np.ascontiguousarray(a, np.float32)
    u = a32.view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32).astype(a.dtype if hasattr(a, "dtype") else np.float32)


_ROUND_POOL = None
_ROUND_CHUNK = 1 << 21          # elements per task: 8 MB of fp32, L2-sized passes


def _pool():
    """Worker threads for the element-wise passes over [B, W] tensors (NumPy releases the GIL in them).  Chunk
    boundaries depend on the array shape only, never on the number of threads: results do not depend on the machine."""
    global _ROUND_POOL
    if _ROUND_POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        _ROUND_POOL = ThreadPoolExecutor(max_workers=max(1, min(16, cores)))
    return _ROUND_POOL


def _row_chunks(nrows, width):
    rows = max(1, _ROUND_CHUNK // max(1, int(width)))
    return [(lo, min(nrows, lo + rows)) for lo in range(0, nrows, rows)]


def _par(fn, chunks):
    """[fn(lo, hi) for (lo, hi) in chunks], on the pool when there is more than one chunk (fn must not use the pool)."""
    if len(chunks) == 1:
        return [fn(*chunks[0])]
    jobs = [_pool().submit(fn, lo, hi) for lo, hi in chunks]
    return [j.result() for j in jobs]


def _round_chunk(src, dst):
    """dst (fp32 or fp64, flat) <- src (any float, flat) rounded to bf16.  Same integer arithmetic as
    _round_bf16_reference in 32 bits (the sum wraps exactly where the 64-bit form is truncated by its final cast)."""
    f = src.astype(np.float32)          # (a copy even when src is fp32: rounded in place below)
    u = f.view(np.uint32)
    r = u >> 16
    r &= 1
    r += 0x7FFF
    u += r
    u &= 0xFFFF0000
    dst[...] = f


def _round_serial(a):
    """round_bf16 on the calling thread (for use inside a _par task)."""
    src = np.ascontiguousarray(a)
    out = np.empty(src.shape, src.dtype)
    _round_chunk(src.reshape(-1), out.reshape(-1))
    return out


def round_bf16(a):
    """Round to bfloat16 (nearest even), result in the dtype of ``a``.  The tests call this on [16384, 2048] fp64
    tensors ninety times per oracle step: chunks of the flat array are rounded by a small thread pool, 26 s -> 3 s
    per step at 4 x 1024, B = 8192; bit-identical to _round_bf16_reference (tests/test_oracle_golden.py)."""
    dt = a.dtype if hasattr(a, "dtype") else np.float32
    if hasattr(a, "flags") and a.ndim == 2 and not a.flags.c_contiguous and a.T.flags.c_contiguous:
        return round_bf16(a.T).T        # (a transposed view: round the array behind it, no strided copy)
    src = np.ascontiguousarray(a)
    if src.dtype not in (np.float32, np.float64):
        src = src.astype(np.float32)
    out = np.empty(src.shape, dt)
    fs, fo = src.reshape(-1), out.reshape(-1)
    n = fs.size
    if n <= _ROUND_CHUNK:
        _round_chunk(fs, fo)
        return out
    jobs = [_pool().submit(_round_chunk, fs[i:i + _ROUND_CHUNK], fo[i:i + _ROUND_CHUNK])
            for i in range(0, n, _ROUND_CHUNK)]
    for j in jobs:
        j.result()
    return out


def _mm(a, b):
    """a @ b with the configured operand rounding."""
    if _GEMM_ROUND in ("bf16", "bf16s"):
        return round_bf16(a) @ round_bf16(b)
    return a @ b


def _st(a):
    """A tensor as the build stores it: rounded to bf16 in "bf16s" mode, unchanged otherwise."""
    return round_bf16(a) if _GEMM_ROUND == "bf16s" else a


# --------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------
def _heavy_fwd(st, h, a_in, mask, training, dtype, update_running, momentum):
    """heavy_linear, model/bilinear.py:7-13:
    Linear (z = a W^T + b) -> BatchNorm1d -> ReLU -> Dropout(0.5).

    BatchNorm1d (training): normalise with the biased batch variance; update
    running_mean <- (1-m) rm + m mu, running_var <- (1-m) rv + m var*B/(B-1)
    (unbiased), num_batches_tracked += 1.  ``momentum=None`` is PyTorch's
    cumulative-average mode (what reset_statistics, model/bilinear.py:43-55,
    selects): factor = 1/num_batches_tracked.
    Dropout (training): a = y * mask / (1-p), values exactly {0, 2y}."""
    W = st[h + ".0.weight"].astype(dtype)
    b = st[h + ".0.bias"].astype(dtype)
    gamma = st[h + ".1.weight"].astype(dtype)
    beta = st[h + ".1.bias"].astype(dtype)
    z = _mm(a_in, W.T)
    n = z.shape[0]
    chunks = _row_chunks(n, z.shape[1])        # row blocks of the [B, W] passes below (one block for small tensors)

    def add_bias(lo, hi):
        z[lo:hi] += b
    _par(add_bias, chunks)
    if training:
        mu = sum(_par(lambda lo, hi: z[lo:hi].sum(axis=0, dtype=np.float64), chunks)) / n

        def sq_dev(lo, hi):
            d = z[lo:hi].astype(np.float64) - mu
            return (d * d).sum(axis=0)
        var = sum(_par(sq_dev, chunks)) / n     # biased
        if update_running:
            nbt = int(st[h + ".1.num_batches_tracked"]) + 1
            st[h + ".1.num_batches_tracked"] = np.asarray(nbt, np.int64)
            f = (1.0 / nbt) if momentum is None else momentum
            unbiased = var * (n / max(n - 1, 1))
            st[h + ".1.running_mean"] = ((1 - f) * st[h + ".1.running_mean"].astype(np.float64)
                                         + f * mu).astype(st[h + ".1.running_mean"].dtype)
            st[h + ".1.running_var"] = ((1 - f) * st[h + ".1.running_var"].astype(np.float64)
                                        + f * unbiased).astype(st[h + ".1.running_var"].dtype)
        mu = mu.astype(dtype)
        invstd = (1.0 / np.sqrt(var + BN_EPS)).astype(dtype)
    else:
        mu = st[h + ".1.running_mean"].astype(dtype)
        invstd = (1.0 / np.sqrt(st[h + ".1.running_var"].astype(np.float64) + BN_EPS)).astype(dtype)
    zhat, y, a = np.empty_like(z), np.empty_like(z), np.empty_like(z)
    keep = np.empty_like(z) if training else None
    rounding = _GEMM_ROUND == "bf16s"
    two = dtype(1.0 / (1.0 - DROPOUT_P))

    def tail(lo, hi):
        # ("bf16s": z is stored rounded, AFTER the statistics above were taken from the un-rounded values)
        zc = _round_serial(z[lo:hi]) if rounding else z[lo:hi]
        if rounding:
            z[lo:hi] = zc
        zh = (zc - mu) * invstd
        yc = zh * gamma + beta
        r = np.maximum(yc, 0)
        zhat[lo:hi] = zh
        y[lo:hi] = yc
        if training:
            kc = mask[lo:hi].astype(dtype)
            keep[lo:hi] = kc
            a[lo:hi] = r * kc * two
        else:
            a[lo:hi] = r
    _par(tail, chunks)
    cache = dict(a_in=a_in, z=z, zhat=zhat, invstd=invstd, y=y, keep=keep)
    return a.astype(dtype), cache


def forward(st, x, masks=None, training=True, dtype=np.float32,
            update_running=True, momentum=BN_MOMENTUM):
    """BilinearUnit.forward, model/bilinear.py:31-41:
    encode -> for each block: skip = a; a = block(a); a = a + skip -> decode.
    Returns (prediction [B,48], cache for backward).  ``st`` running stats are
    updated in place when training (as nn.BatchNorm1d does)."""
    dtype = np.dtype(dtype).type
    num_blocks = (sum(1 for k in st if k.endswith(".0.weight")) - 1) // 2
    names = heavy_names(num_blocks)
    a = _st(np.asarray(x, dtype))
    caches = []
    a, c = _heavy_fwd(st, names[0], a, None if masks is None else masks[0],
                      training, dtype, update_running, momentum)
    a = _st(a)
    caches.append(c)
    li = 1
    for _ in range(num_blocks):
        skip = a
        for _l in range(2):
            a, c = _heavy_fwd(st, names[li], a, None if masks is None else masks[li],
                              training, dtype, update_running, momentum)
            if _l == 0:
                a = _st(a)
            caches.append(c)
            li += 1
        a = _st(a + skip)                              # model/bilinear.py:38
    Wd = st["decode.weight"].astype(dtype)
    bd = st["decode.bias"].astype(dtype)
    pred = _mm(a, Wd.T) + bd                           # model/bilinear.py:39
    return pred, dict(layers=caches, a_last=a, names=names, num_blocks=num_blocks)


# --------------------------------------------------------------------------
# loss
# --------------------------------------------------------------------------
def mse_loss(pred, target):
    """nn.MSELoss() (mean reduction), train_bilinear.py:49,78:
    loss = sum((p-t)^2)/(B*48); d loss/d p = 2 (p-t)/(B*48)."""
    diff = pred - target
    n = diff.size
    loss = float((diff.astype(np.float64) ** 2).sum() / n)
    dpred = (diff * pred.dtype.type(2.0 / n)).astype(pred.dtype)
    return loss, dpred


# --------------------------------------------------------------------------
# backward  (autograd of forward + mse, train_bilinear.py:79)
# --------------------------------------------------------------------------
def _heavy_bwd(st, h, c, d_a, dtype, need_dx=True):
    """Backward of heavy_linear.
    dropout+relu: dY = dA * keep/(1-p) * [y>0]
    batchnorm   : dgamma = sum_B dY*zhat ; dbeta = sum_B dY ;
                  dZ = gamma*invstd * (dY - dbeta/B - zhat*dgamma/B)
    linear      : dW = dZ^T a_in ; db = sum_B dZ ; dA_in = dZ W."""
    gamma = st[h + ".1.weight"].astype(dtype)
    W = st[h + ".0.weight"].astype(dtype)
    n = d_a.shape[0]
    chunks = _row_chunks(n, d_a.shape[1])
    keep_t, y_t, zhat_t = c["keep"], c["y"], c["zhat"]
    dY = np.empty(d_a.shape, dtype)
    two = dtype(1.0 / (1.0 - DROPOUT_P))

    def gate(lo, hi):
        dy = d_a[lo:hi] * keep_t[lo:hi].astype(dtype) * two * (y_t[lo:hi] > 0)
        dY[lo:hi] = dy
        d64 = dy.astype(np.float64)
        return (d64 * zhat_t[lo:hi]).sum(axis=0), d64.sum(axis=0)
    parts = _par(gate, chunks)
    dgamma = sum(p[0] for p in parts)
    dbeta = sum(p[1] for p in parts)
    gi = gamma * c["invstd"]
    c1, c2 = (dbeta / n).astype(dtype), (dgamma / n).astype(dtype)
    dZ = np.empty(d_a.shape, dtype)
    rounding = _GEMM_ROUND == "bf16s"

    def bn_bwd(lo, hi):
        dz = (gi * (dY[lo:hi] - c1 - zhat_t[lo:hi] * c2)).astype(dtype)
        dZ[lo:hi] = _round_serial(dz) if rounding else dz
    _par(bn_bwd, chunks)
    g = {
        h + ".0.weight": _mm(dZ.T, c["a_in"]),
        h + ".0.bias": dZ.sum(axis=0, dtype=np.float64).astype(dtype),
        h + ".1.weight": dgamma.astype(dtype),
        h + ".1.bias": dbeta.astype(dtype),
    }
    d_in = _mm(dZ, W) if need_dx else None
    return d_in, g


def backward(st, cache, dpred, dtype=np.float32):
    """Gradients of every parameter (dict keyed like state_dict).  The network
    input gets no gradient (train_bilinear.py:72)."""
    dtype = np.dtype(dtype).type
    names, nb = cache["names"], cache["num_blocks"]
    grads = {}
    Wd = st["decode.weight"].astype(dtype)
    grads["decode.weight"] = _mm(dpred.T, cache["a_last"])
    grads["decode.bias"] = dpred.sum(axis=0, dtype=np.float64).astype(dtype)
    d_a = _st(_mm(dpred, Wd))
    li = len(names) - 1
    for _ in range(nb):
        d_skip = d_a                                    # a = block(a) + skip
        for _l in range(2):
            d_a, g = _heavy_bwd(st, names[li], cache["layers"][li], d_a, dtype)
            if _l == 0:
                d_a = _st(d_a)
            grads.update(g)
            li -= 1
        d_a = _st(d_a + d_skip)
    _, g = _heavy_bwd(st, names[0], cache["layers"][0], d_a, dtype, need_dx=False)
    grads.update(g)
    return grads


# --------------------------------------------------------------------------
# clip + Adam + lr decay  (train_bilinear.py:66-70, 81, 83)
# --------------------------------------------------------------------------
def clip_grad_norm(grads, keys, max_norm=CLIP_MAX_NORM):
    """nn.utils.clip_grad_norm_(params, max_norm=1), train_bilinear.py:81:
    total = ||all grads||_2 ; coef = min(1, max_norm/(total+1e-6)) ; g *= coef
    (PyTorch multiplies unconditionally by the clamped coefficient)."""
    tot = 0.0
    for k in keys:
        tot += float((grads[k].astype(np.float64) ** 2).sum())
    total_norm = math.sqrt(tot)
    coef = min(1.0, max_norm / (total_norm + 1e-6))
    for k in keys:
        grads[k] = (grads[k] * grads[k].dtype.type(coef)).astype(grads[k].dtype)
    return total_norm, coef


def adam_init(st, keys):
    return dict(step=0,
                exp_avg={k: np.zeros_like(st[k]) for k in keys},
                exp_avg_sq={k: np.zeros_like(st[k]) for k in keys})


def adam_step(st, grads, opt, keys, lr, betas=ADAM_BETAS, eps=ADAM_EPS):
    """torch.optim.Adam (lr given, betas (0.9,0.999), eps 1e-8, no weight decay,
    no amsgrad; model/bilinear.py:60, train_bilinear.py:83):
      m <- m + (1-b1)(g - m)        (lerp)
      v <- b2 v + (1-b2) g*g
      p <- p - (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)"""
    opt["step"] += 1
    t = opt["step"]
    b1, b2 = betas
    bc1 = 1.0 - b1 ** t
    bc2 = 1.0 - b2 ** t
    step_size = lr / bc1
    bc2_sqrt = math.sqrt(bc2)
    for k in keys:
        g = grads[k]
        dt = g.dtype.type
        m = opt["exp_avg"][k]
        v = opt["exp_avg_sq"][k]
        m += (g - m) * dt(1.0 - b1)
        v *= dt(b2)
        v += (g * g) * dt(1.0 - b2)
        denom = np.sqrt(v) / dt(bc2_sqrt) + dt(eps)
        st[k] = (st[k] - dt(step_size) * (m / denom)).astype(st[k].dtype)


def lr_decay_condition(step):
    """util/config.py:21."""
    return step % 100000 == 0 or step == 1


def lr_decay_function(step):
    """util/config.py:22."""
    return 1.0e-3 * 0.96 ** (step / 100000)


def train_step(st, opt, x, t, masks, lr, dtype=np.float32):
    """The step body of train_bilinear.py:75-83 (zero_grad, forward, MSE,
    backward, clip_grad_norm_(1), Adam).  Returns a dict of observables."""
    keys = [k for k in st if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]
    pred, cache = forward(st, x, masks, training=True, dtype=dtype)
    loss, dpred = mse_loss(pred, np.asarray(t, pred.dtype))
    grads = backward(st, cache, dpred, dtype=dtype)
    raw = {k: v.copy() for k, v in grads.items()}
    total_norm, coef = clip_grad_norm(grads, keys)
    adam_step(st, grads, opt, keys, lr)
    return dict(pred=pred, loss=loss, grads_raw=raw, grads=grads,
                total_norm=total_norm, clip_coef=coef)


# --------------------------------------------------------------------------
# "next" row: MPJPE of valid_bilinear.py:53-60,76-83
# --------------------------------------------------------------------------
def mpjpe_sum(pred, target, mean, stddev):
    """valid_bilinear.py:53-60: de-normalise both with the train-set mean/std,
    view as [B,16,3], per-sample sum over joints of the Euclidean distance."""
    p = (stddev * pred + mean).reshape(-1, NUM_JOINT, 3)
    g = (stddev * target + mean).reshape(-1, NUM_JOINT, 3)
    return np.sqrt(((p - g) ** 2).sum(axis=2)).sum(axis=1)


# ----------------------------------------------------------------------------
# Input pipeline: the preprocessing the reference's H36M.Dataset does in
# its constructor and in __getitem__, restated on whole arrays.
# ----------------------------------------------------------------------------
def h36m_decode_action(image_name):
    """Action of one sample from its image name, /root/reference/H36M/util.py:13-22
    ('S1_Directions_1.54138969_000001.jpg' -> 'Directions_1'), then the validator's merge of the
    two sub-actions, /root/reference/valid_bilinear.py:64 ('Directions_1' -> 'Directions')."""
    subject_action = image_name.split('.')[0]
    parts = subject_action.split('_')
    action = parts[1]
    if len(parts) >= 3:
        action = action + '_' + parts[2]
    return action.split('_')[0]


def h36m_flatten(part, S):
    """Raw annotations -> flat features, /root/reference/H36M/data.py:36-59:
    part [n,17,2] loses joint 9 (nose) -> [n,32]; S [n,17,3] is root-centred (joint 0) and loses
    the pelvis -> [n,48].  float32 like the reference."""
    part = np.asarray(part, dtype=np.float32)
    S = np.asarray(S, dtype=np.float32)
    p = np.delete(part, 9, axis=1)
    s = (S - S[:, 0:1, :])[:, 1:, :]
    return p.reshape(-1, 32), s.reshape(-1, 48)


def h36m_stats(flat):
    """Per-feature mean and (population) standard deviation, H36M/data.py:58-59 (np.mean / np.std)."""
    return np.mean(flat, axis=0), np.std(flat, axis=0)


def h36m_normalise(flat, mean, stddev):
    """__getitem__ z-scoring with the TRAIN statistics, H36M/data.py:108-110."""
    return (flat - mean) / stddev
